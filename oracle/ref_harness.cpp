/*
 * TEST INFRASTRUCTURE — reference harness ("oracle/_ref").
 *
 * This file is the only translation unit of ours that touches the reference's
 * sources, and it does so by *including them where they lie* under
 * /root/reference (see oracle/Makefile); nothing of the reference is copied into
 * this repository.  It exists only in the build container: the resulting
 * oracle/_ref/libelas_ref.so travels to the GPU box as a built artefact, the
 * reference sources never do.
 *
 * What it does: runs the reference's serial LIBELAS (`Elas::process`,
 * src/serial_includes/elas/elas.cpp:31-150) stage by stage, calling the
 * reference's own member functions in the reference's own order, and keeps a
 * copy of every intermediate so tests can (a) pin the CPU restatement in
 * oracle/elas_oracle.cpp and (b) generate tests/golden/ fixtures.
 *
 * Canonical mode: the reference reads uninitialised heap (SURVEY.md §0 fact 5:
 * descriptor.cpp:31 border never written, elas.cpp:1308 D_tmp border).  With
 * glibc's M_PERTURB=255 every malloc'd byte is 0x00, which equals the first call
 * in a fresh process.  `canonical != 0` turns that on for the duration of a call.
 */
#include <malloc.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "elas_params.h"

/* Reach the stage functions: all of them are `private` in elas.h:164-265. */
#define private public
#include "elas.cpp" /* -I<reference>/src/serial_includes/elas ; pulls in elas.h, descriptor.h, matrix.h, triangle.h */
#undef private

namespace {

std::map<std::string, std::vector<uint8_t>> g_store;

template <class T>
void put(const std::string &name, const T *data, size_t count) {
    std::vector<uint8_t> &v = g_store[name];
    v.resize(count * sizeof(T));
    if (count)
        memcpy(v.data(), data, count * sizeof(T));
}

Elas::parameters to_ref(const elas_params *p) {
    Elas::parameters q(Elas::ROBOTICS);
    q.disp_min = p->disp_min;
    q.disp_max = p->disp_max;
    q.support_threshold = p->support_threshold;
    q.support_texture = p->support_texture;
    q.candidate_stepsize = p->candidate_stepsize;
    q.incon_window_size = p->incon_window_size;
    q.incon_threshold = p->incon_threshold;
    q.incon_min_support = p->incon_min_support;
    q.add_corners = p->add_corners != 0;
    q.grid_size = p->grid_size;
    q.beta = p->beta;
    q.gamma = p->gamma;
    q.sigma = p->sigma;
    q.sradius = p->sradius;
    q.match_texture = p->match_texture;
    q.lr_threshold = p->lr_threshold;
    q.speckle_sim_threshold = p->speckle_sim_threshold;
    q.speckle_size = p->speckle_size;
    q.ipol_gap_width = p->ipol_gap_width;
    q.filter_median = p->filter_median != 0;
    q.filter_adaptive_mean = p->filter_adaptive_mean != 0;
    q.postprocess_only_left = p->postprocess_only_left != 0;
    q.subsampling = p->subsampling != 0;
    return q;
}

struct PerturbGuard {
    bool on;
    explicit PerturbGuard(bool enable) : on(enable) {
        if (on)
            mallopt(M_PERTURB, 255);
    }
    ~PerturbGuard() {
        if (on)
            mallopt(M_PERTURB, 0);
    }
};

void put_tris(const std::string &idx_name, const std::string &plane_name, const std::vector<Elas::triangle> &tri) {
    std::vector<int32_t> idx(tri.size() * 3);
    std::vector<float> pl(tri.size() * 6);
    for (size_t i = 0; i < tri.size(); i++) {
        idx[3 * i + 0] = tri[i].c1;
        idx[3 * i + 1] = tri[i].c2;
        idx[3 * i + 2] = tri[i].c3;
        pl[6 * i + 0] = tri[i].t1a;
        pl[6 * i + 1] = tri[i].t1b;
        pl[6 * i + 2] = tri[i].t1c;
        pl[6 * i + 3] = tri[i].t2a;
        pl[6 * i + 4] = tri[i].t2b;
        pl[6 * i + 5] = tri[i].t2c;
    }
    put(idx_name, idx.data(), idx.size());
    put(plane_name, pl.data(), pl.size());
}

}  // namespace

extern "C" {

/* Straight call of the reference operator seam (elas.h:162).  Returns seconds spent inside
 * Elas::process (averaged over `reps` calls) so bench.py can time the real reference. */
double ref_process(const elas_params *p, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride, float *D1, float *D2, int canonical, int reps) {
    PerturbGuard guard(canonical != 0);
    Elas elas(to_ref(p));
    const int32_t dims[3] = {W, H, stride};
    if (reps < 1)
        reps = 1;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++)
        elas.process(const_cast<uint8_t *>(I1), const_cast<uint8_t *>(I2), D1, D2, dims);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count() / reps;
}

/* Stage-by-stage run.  Mirrors the statement order of Elas::process (elas.cpp:31-150) but keeps
 * every intermediate.  Returns the number of support points, or <0 on error. */
int ref_run_stages(const elas_params *p, const uint8_t *I1_, const uint8_t *I2_, int W, int H, int stride) {
    g_store.clear();
    const bool sub = p->subsampling != 0;
    PerturbGuard guard(true);
    Elas elas(to_ref(p));
    elas.width = W;
    elas.height = H;
    elas.bpl = W + 15 - (W - 1) % 16; /* elas.cpp:35 */
    const int bpl = elas.bpl;

    /* elas.cpp:38-50 */
    elas.I1 = (uint8_t *)_mm_malloc(bpl * H, 16);
    elas.I2 = (uint8_t *)_mm_malloc(bpl * H, 16);
    memset(elas.I1, 0, bpl * H);
    memset(elas.I2, 0, bpl * H);
    for (int v = 0; v < H; v++) {
        memcpy(elas.I1 + v * bpl, I1_ + v * stride, W);
        memcpy(elas.I2 + v * bpl, I2_ + v * stride, W);
    }

    int n_support = 0;
    {
        /* elas.cpp:55-56 */
        Descriptor desc1(elas.I1, W, H, bpl, sub);
        Descriptor desc2(elas.I2, W, H, bpl, sub);
        put("desc1", desc1.I_desc, (size_t)16 * W * H);
        put("desc2", desc2.I_desc, (size_t)16 * W * H);

        /* Extra: the raw candidate lattice before the in-place filters, produced by the reference's own
         * computeMatchingDisparity in the loop order of elas.cpp:394-411. */
        {
            int step = p->candidate_stepsize;
            if (sub) step += step % 2; /* elas.cpp:377-378 */
            int Wc = 0, Hc = 0;
            for (int u = 0; u < W; u += step) Wc++;
            for (int v = 0; v < H; v += step) Hc++;
            std::vector<int16_t> dcan((size_t)Wc * Hc, 0);
            for (int uc = 1; uc < Wc; uc++) {
                for (int vc = 1; vc < Hc; vc++) {
                    int32_t u = uc * step, v = vc * step;
                    bool f = false, t = true;
                    int16_t out = -1;
                    int16_t d = elas.computeMatchingDisparity(u, v, desc1.I_desc, desc2.I_desc, f);
                    if (d >= 0) {
                        int32_t u2 = u - d;
                        int16_t d2 = elas.computeMatchingDisparity(u2, v, desc1.I_desc, desc2.I_desc, t);
                        if (d2 >= 0 && abs(d - d2) <= p->lr_threshold)
                            out = d;
                    }
                    dcan[(size_t)vc * Wc + uc] = out;
                }
            }
            put("dcan_raw", dcan.data(), dcan.size());
            int32_t dims[2] = {Wc, Hc};
            put("dcan_dims", dims, 2);
        }

        /* elas.cpp:61 */
        std::vector<Elas::support_pt> p_support = elas.computeSupportMatches(desc1.I_desc, desc2.I_desc);
        n_support = (int)p_support.size();
        {
            std::vector<int32_t> s(p_support.size() * 3);
            for (size_t i = 0; i < p_support.size(); i++) {
                s[3 * i] = p_support[i].u;
                s[3 * i + 1] = p_support[i].v;
                s[3 * i + 2] = p_support[i].d;
            }
            put("support", s.data(), s.size());
        }
        if (p_support.size() < 3) { /* elas.cpp:64-69 */
            _mm_free(elas.I1);
            _mm_free(elas.I2);
            return n_support;
        }

        /* elas.cpp:74-81 */
        std::vector<Elas::triangle> tri_1 = elas.computeDelaunayTriangulation(p_support, 0);
        std::vector<Elas::triangle> tri_2 = elas.computeDelaunayTriangulation(p_support, 1);
        elas.computeDisparityPlanes(p_support, tri_1, 0);
        elas.computeDisparityPlanes(p_support, tri_2, 1);
        put_tris("tri1", "planes1", tri_1);
        put_tris("tri2", "planes2", tri_2);

        /* elas.cpp:88-95 */
        int32_t grid_width = (int32_t)ceil((float)W / (float)p->grid_size);
        int32_t grid_height = (int32_t)ceil((float)H / (float)p->grid_size);
        int32_t grid_dims[3] = {p->disp_max + 2, grid_width, grid_height};
        size_t gcount = (size_t)(p->disp_max + 2) * grid_height * grid_width;
        int32_t *g1 = (int32_t *)calloc(gcount, sizeof(int32_t));
        int32_t *g2 = (int32_t *)calloc(gcount, sizeof(int32_t));
        elas.createGrid(p_support, g1, grid_dims, 0);
        elas.createGrid(p_support, g2, grid_dims, 1);
        put("grid1", g1, gcount);
        put("grid2", g2, gcount);
        put("grid_dims", grid_dims, 3);

        /* The driver hands in zero-initialised maps (stereo_vision.cpp:304-305). */
        const size_t NM = sub ? (size_t)(W / 2) * (H / 2) : (size_t)W * H; /* elas.h:160-161: half-size maps when subsampling */
        std::vector<float> D1(NM, 0.f), D2(NM, 0.f);

        /* elas.cpp:100-101 */
        elas.computeDisparity(p_support, tri_1, g1, grid_dims, desc1.I_desc, desc2.I_desc, 0, D1.data());
        elas.computeDisparity(p_support, tri_2, g2, grid_dims, desc1.I_desc, desc2.I_desc, 1, D2.data());
        put("wta1", D1.data(), D1.size());
        put("wta2", D2.data(), D2.size());

        /* elas.cpp:106 */
        elas.leftRightConsistencyCheck(D1.data(), D2.data());
        put("lr1", D1.data(), D1.size());
        put("lr2", D2.data(), D2.size());

        /* elas.cpp:111-113 */
        elas.removeSmallSegments(D1.data());
        if (!p->postprocess_only_left)
            elas.removeSmallSegments(D2.data());
        put("speckle1", D1.data(), D1.size());
        put("speckle2", D2.data(), D2.size());

        /* elas.cpp:118-120 */
        elas.gapInterpolation(D1.data());
        if (!p->postprocess_only_left)
            elas.gapInterpolation(D2.data());
        put("gap1", D1.data(), D1.size());
        put("gap2", D2.data(), D2.size());

        /* elas.cpp:122-129 */
        if (p->filter_adaptive_mean) {
            elas.adaptiveMean(D1.data());
            if (!p->postprocess_only_left)
                elas.adaptiveMean(D2.data());
        }
        put("amean1", D1.data(), D1.size());
        put("amean2", D2.data(), D2.size());

        /* elas.cpp:131-138 */
        if (p->filter_median) {
            elas.median(D1.data());
            if (!p->postprocess_only_left)
                elas.median(D2.data());
        }
        put("final1", D1.data(), D1.size());
        put("final2", D2.data(), D2.size());

        free(g1);
        free(g2);
    }
    _mm_free(elas.I1);
    _mm_free(elas.I2);
    return n_support;
}

/* Delaunay stage alone: points are (x,y) float pairs exactly as elas.cpp:449-461 builds them;
 * calls the vendored Triangle with the reference's switches "zQB" (elas.cpp:483-484).
 * Writes up to cap triangles (3 ints each); returns the triangle count. */
int ref_delaunay(const float *xy, int n, int32_t *tri_out, int cap) {
    struct triangulateio in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.numberofpoints = n;
    in.pointlist = (float *)malloc(sizeof(float) * 2 * n);
    memcpy(in.pointlist, xy, sizeof(float) * 2 * n);
    char parameters[] = "zQB";
    triangulate(parameters, &in, &out, NULL);
    int nt = out.numberoftriangles;
    for (int i = 0; i < nt && i < cap; i++) {
        tri_out[3 * i] = out.trianglelist[3 * i];
        tri_out[3 * i + 1] = out.trianglelist[3 * i + 1];
        tri_out[3 * i + 2] = out.trianglelist[3 * i + 2];
    }
    free(in.pointlist);
    free(out.pointlist);
    free(out.trianglelist);
    return nt;
}

long ref_size(const char *name) {
    auto it = g_store.find(name);
    return it == g_store.end() ? -1 : (long)it->second.size();
}

long ref_get(const char *name, void *out, long cap) {
    auto it = g_store.find(name);
    if (it == g_store.end())
        return -1;
    long n = (long)it->second.size();
    if (n > cap)
        return -2;
    memcpy(out, it->second.data(), n);
    return n;
}

} /* extern "C" */
