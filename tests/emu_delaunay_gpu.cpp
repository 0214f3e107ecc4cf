// CPU emulation of csrc/delaunay_gpu.hip: the device functions compiled as plain C++, the tree processed depth by depth with
// the nodes of one depth in arbitrary (here: reversed) order, against Delaunay::triangulate.  tests/test_sanitizers.py builds it
// with AddressSanitizer; every loop of the merge gets a step budget so that a logic error ends the run instead of hanging.
#define DG_HOST_EMULATION 1
#define __device__
#define __host__
#define __forceinline__ inline
#include "delaunay_gpu.hip"
#include "host_stage.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

int main() {
    std::mt19937 rng(11);
    int bad = 0;
    for (int it = 0; it < 400; it++) {
        int n = 3 + rng() % (it % 10 == 0 ? 3900 : 400);
        std::vector<int32_t> xy(2 * n);
        for (int i = 0; i < n; i++) {
            if (it % 3 == 0) { xy[2 * i] = (int)(rng() % 250) * 5; xy[2 * i + 1] = (int)(rng() % 75) * 5; }
            else if (it % 3 == 1) { xy[2 * i] = (int)(rng() % 1300) - 50; xy[2 * i + 1] = (int)(rng() % 75) * 5; }
            else { xy[2 * i] = (int)(rng() % 12) * 5; xy[2 * i + 1] = (int)(rng() % 12) * 5; }
        }
        sv::Delaunay dl;
        std::vector<int32_t> want(6 * n + 24), ids(n);
        const int nw = dl.triangulate(xy.data(), n, want.data(), 2 * n + 8);
        sv::Delaunay dl2;
        const int m = dl2.kd_ordered_ids(xy.data(), n, ids.data());
        if (m < 3) { if (nw != 0) bad++; continue; }
        const int nslots = 2 * m - 1;
        std::vector<uint32_t> res(2 << 12, 0);
        std::vector<sv::dg::DTri> T(nslots);
        std::vector<uint32_t> pxy(n);
        std::vector<uint16_t> ord(m);
        for (int i = 0; i < n; i++) pxy[i] = ((uint32_t)xy[2 * i] & 0xFFFFu) | ((uint32_t)xy[2 * i + 1] << 16);
        for (int i = 0; i < m; i++) ord[i] = (uint16_t)ids[i];
        T[0].w[0] = 0, T[0].w[1] = 0xFFFFu << 16, T[0].w[2] = 0xFFFFFFFFu;
        const sv::dg::Mesh M{&T[0].w[0], pxy.data()};
        const uint16_t *F = reinterpret_cast<const uint16_t *>(T.data());
        for (int d = 12; d >= 0; d--)
            for (int j = (1 << d) - 1; j >= 0; j--) {
                if (it % 2)  // the engine's sets are narrow (32-bit in-circle terms); both forms must give the same mesh on them
                    sv::dg::d_process_node<true>(M, res.data(), ord.data(), m, d, j);
                else
                    sv::dg::d_process_node<false>(M, res.data(), ord.data(), m, d, j);
            }
        std::vector<int32_t> got;
        for (int t = 1; t < nslots; t++) {
            if (F[6 * t + 3] == 0xFFFF || F[6 * t + 4] == 0xFFFF || F[6 * t + 5] == 0xFFFF) continue;
            got.push_back(F[6 * t + 4]); got.push_back(F[6 * t + 5]); got.push_back(F[6 * t + 3]);
        }
        if ((int)got.size() != 3 * nw || memcmp(got.data(), want.data(), sizeof(int32_t) * got.size())) {
            bad++;
            if (bad < 5) printf("mismatch: case %d n %d m %d  tris %d vs %zu\n", it, n, m, nw, got.size() / 3);
        }
    }
    // Sets cut into subtrees (dg_subtree / dg_top of the kernels, restated sequentially): every subtree in a local 16-bit mesh
    // with local vertex numbers, exported into one 32-bit mesh, then the merges above the cut; subtree sizes from 7 to 4000.
    int cut_cases = 0, deepest = 0;
    for (int it = 0; it < 120; it++) {
        const int sub_max = it % 4 == 0 ? 4000 : 7 + (int)(rng() % 600);
        int n = sub_max + 1 + (int)(rng() % (it % 4 == 0 ? 30000 : 20 * sub_max));
        if (sv::dg::dg_cut_depth(n, sub_max) > sv::dg::DG_CUT_MAX) n = sub_max << sv::dg::DG_CUT_MAX;
        std::vector<int32_t> xy(2 * n);
        for (int i = 0; i < n; i++) {
            if (it % 3 == 0) { xy[2 * i] = (int)(rng() % 768) * 5; xy[2 * i + 1] = (int)(rng() % 432) * 5; }
            else if (it % 3 == 1) { xy[2 * i] = (int)(rng() % 4000) - 190; xy[2 * i + 1] = (int)(rng() % 432) * 5; }
            else { xy[2 * i] = (int)(rng() % 60) * 5; xy[2 * i + 1] = (int)(rng() % 60) * 5; }
        }
        sv::Delaunay dl;
        std::vector<int32_t> want(6 * (size_t)n + 24), ids(n);
        const int nw = dl.triangulate(xy.data(), n, want.data(), 2 * n + 8);
        sv::Delaunay dl2;
        const int m = dl2.kd_ordered_ids(xy.data(), n, ids.data());
        if (m <= sub_max) continue;  // (many duplicates: the set fits LDS after all - the first loop's case)
        const int c = sv::dg::dg_cut_depth(m, sub_max);
        cut_cases++;
        deepest = c > deepest ? c : deepest;
        std::vector<sv::dg::GTri> G(2 * (size_t)m);
        std::vector<uint32_t> gres(4 << sv::dg::DG_CUT_MAX, 0);
        for (int k = 0; k < 3; k++) { G[0].w[k] = 0; G[0].w[3 + k] = sv::dg::GHOST32; }
        for (int j = (1 << c) - 1; j >= 0; j--) {
            int lo, ns, axis0;
            uint32_t slot0;
            if (!sv::dg::d_node(m, c, j, lo, ns, slot0, axis0)) { bad++; continue; }
            const int depth = sv::dg::dg_depth(ns);
            std::vector<uint32_t> res(2 << depth, 0);
            std::vector<sv::dg::DTri> T(2 * ns - 1);
            std::vector<uint32_t> pxy(ns);
            std::vector<uint16_t> ord(ns);
            for (int i = 0; i < ns; i++) { pxy[i] = ((uint32_t)xy[2 * ids[lo + i]] & 0xFFFFu) | ((uint32_t)xy[2 * ids[lo + i] + 1] << 16); ord[i] = (uint16_t)i; }
            T[0].w[0] = 0, T[0].w[1] = 0xFFFFu << 16, T[0].w[2] = 0xFFFFFFFFu;
            const sv::dg::Mesh M{&T[0].w[0], pxy.data()};
            const uint16_t *F = reinterpret_cast<const uint16_t *>(T.data());
            for (int d = depth; d >= 0; d--)
                for (int q = (1 << d) - 1; q >= 0; q--) sv::dg::d_process_node<true>(M, res.data(), ord.data(), ns, d, q, axis0);
            for (int t = 1; t < 2 * ns - 1; t++)
                for (int k = 0; k < 3; k++) {
                    G[slot0 + t - 1].w[k] = sv::dg::dg_global_handle(F[6 * t + k], slot0);
                    G[slot0 + t - 1].w[3 + k] = F[6 * t + 3 + k] == 0xFFFF ? sv::dg::GHOST32 : (uint32_t)ids[lo + F[6 * t + 3 + k]];
                }
            gres[2 * ((1 << c) + j)] = sv::dg::dg_global_handle(res[1] & 0xFFFFu, slot0);
            gres[2 * ((1 << c) + j) + 1] = sv::dg::dg_global_handle(res[1] >> 16, slot0);
        }
        const sv::dg::MeshG MG{G.data(), xy.data(), 16u * (uint32_t)m + 4096u};
        for (int d = c - 1; d >= 0; d--)
            for (int j = (1 << d) - 1; j >= 0; j--) sv::dg::dg_top_node<true>(MG, gres.data(), m, d, j);
        std::vector<int32_t> got;
        for (int t = 1; t < 2 * m - 1; t++) {
            if (G[t].w[3] == sv::dg::GHOST32 || G[t].w[4] == sv::dg::GHOST32 || G[t].w[5] == sv::dg::GHOST32) continue;
            got.push_back((int32_t)G[t].w[4]); got.push_back((int32_t)G[t].w[5]); got.push_back((int32_t)G[t].w[3]);
        }
        if ((int)got.size() != 3 * nw || memcmp(got.data(), want.data(), sizeof(int32_t) * got.size())) {
            bad++;
            if (bad < 5) printf("mismatch (cut): case %d n %d m %d sub_max %d  tris %d vs %zu\n", it, n, m, sub_max, nw, got.size() / 3);
        }
    }
    printf("cut sets: %d, deepest cut %d\n", cut_cases, deepest);
    if (cut_cases < 60 || deepest < 5) bad++;
    printf("gpu-delaunay emulation done, mismatches: %d\n", bad);
    return bad != 0;
}
