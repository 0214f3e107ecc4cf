// Divide-and-conquer Delaunay triangulation on the GPU (gfx950), bit-compatible with csrc/host_stage.cpp's Delaunay::build,
// i.e. with Triangle 1.6 "zQB" as the reference uses it (elas.cpp:483-484; common_includes/elas/triangle.cpp:5362-5815).
//
// What makes the sequential recursion parallel without changing its result:
//   * the recursion tree only depends on the number of vertices m (halves n>>1 / n-(n>>1) down to groups of 2 or 3);
//   * a subproblem of n vertices allocates exactly 2n-2 triangle slots (2 per pair, 4 per triple, 2 per merge), so the
//     slot range of every node of the tree is known in advance and the pool comes out in the sequential order whatever
//     the order in which independent nodes are processed;
//   * nodes of one depth are independent.
// One workgroup triangulates one vertex set: the mesh lives in LDS (12-byte triangles with 16-bit handles), the tree is
// processed bottom-up, one lane per node of the current depth, a barrier between depths.
//
// Input: the vertices in the k-d order the reference's alternating cuts leave (host: radix sort, duplicate scan, kd_order).
// Output: the triangle list in pool order without the bounding ("ghost") triangles (triangle.cpp:7449-7500).
#ifndef DG_HOST_EMULATION  // tests/emu_delaunay_gpu.cpp compiles the device functions for the CPU
#include <hip/hip_runtime.h>

#include <atomic>

#include "sv_kernels.h"
#endif
#include <stdint.h>

namespace sv {

namespace dg {

constexpr uint32_t GHOST = 0xFFFFu;
constexpr int DG_THREADS = 256;

struct DTri {
    uint16_t nbr[3];  // neighbour handle across edge o: (slot << 2) | orientation; slot 0 = outer space
    uint16_t vtx[3];  // vertex ids, GHOST = the vertex at infinity of a bounding triangle
};

// The triangle array is accessed through a DG_VOLATILE pointer: hipcc's SLP vectoriser (ROCm 7.2, -O2 and above) fuses the 16-bit
// field accesses of neighbouring statements into wider ones across stores that may hit the same triangle through another handle
// and the merge then walks a stale mesh (reproduced on a 7-vertex set; -fno-slp-vectorize or -O1 give the right mesh).
#ifdef DG_NO_VOLATILE  // experiment: rely on -fno-slp-vectorize instead
#define DG_VOLATILE
#else
#define DG_VOLATILE volatile
#endif
#ifdef DG_HOST_EMULATION
#define DG_LDS
#else
#define DG_LDS __attribute__((address_space(3)))  // explicit LDS pointers: ds_* instructions instead of flat_* ones
#endif
struct Mesh {
    DG_LDS DG_VOLATILE DTri *T;
    DG_LDS const int16_t *px, *py;
};

__device__ __forceinline__ uint32_t next3(uint32_t o) { return (9u >> (2 * o)) & 3u; }
__device__ __forceinline__ uint32_t prev3(uint32_t o) { return (18u >> (2 * o)) & 3u; }
__device__ __forceinline__ uint32_t hnext(uint32_t h) { return (h & ~3u) | next3(h & 3u); }
__device__ __forceinline__ uint32_t hprev(uint32_t h) { return (h & ~3u) | prev3(h & 3u); }

#define D_SYM(h) ((uint32_t)M.T[(h) >> 2].nbr[(h)&3u])
#define D_ORG(h) (M.T[(h) >> 2].vtx[next3((h)&3u)])
#define D_DEST(h) (M.T[(h) >> 2].vtx[prev3((h)&3u)])
#define D_APEX(h) (M.T[(h) >> 2].vtx[(h)&3u])
#define D_BOND(a, b)                                  \
    do {                                              \
        const uint32_t a_ = (a), b_ = (b);            \
        M.T[a_ >> 2].nbr[a_ & 3u] = (uint16_t)b_;     \
        M.T[b_ >> 2].nbr[b_ & 3u] = (uint16_t)a_;     \
    } while (0)
#define D_PX(v) ((int64_t)M.px[(v)])
#define D_PY(v) ((int64_t)M.py[(v)])

__device__ __forceinline__ int64_t d_orient(const Mesh &M, uint32_t a, uint32_t b, uint32_t c) {
    return (D_PX(a) - D_PX(c)) * (D_PY(b) - D_PY(c)) - (D_PY(a) - D_PY(c)) * (D_PX(b) - D_PX(c));
}

__device__ __forceinline__ int64_t d_incirc(const Mesh &M, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const int64_t adx = D_PX(a) - D_PX(d), ady = D_PY(a) - D_PY(d);
    const int64_t bdx = D_PX(b) - D_PX(d), bdy = D_PY(b) - D_PY(d);
    const int64_t cdx = D_PX(c) - D_PX(d), cdy = D_PY(c) - D_PY(d);
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
}

// A vertex with its coordinates (the GHOST vertex has none: its coordinates are never used)
struct DV {
    uint32_t id;
    int32_t x, y;
};

__device__ __forceinline__ DV d_vertex(const Mesh &M, uint32_t id) {
    DV v;
    v.id = id;
    const uint32_t safe = id == GHOST ? 0u : id;
    v.x = M.px[safe];
    v.y = M.py[safe];
    return v;
}

__device__ __forceinline__ int64_t dv_orient(const DV &a, const DV &b, const DV &c) {
    return (int64_t)(a.x - c.x) * (b.y - c.y) - (int64_t)(a.y - c.y) * (b.x - c.x);
}

__device__ __forceinline__ int64_t dv_incirc(const DV &a, const DV &b, const DV &c, const DV &d) {
    const int64_t adx = a.x - d.x, ady = a.y - d.y;
    const int64_t bdx = b.x - d.x, bdy = b.y - d.y;
    const int64_t cdx = c.x - d.x, cdy = c.y - d.y;
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
}

__device__ __forceinline__ uint32_t d_make(const Mesh &M, uint32_t slot) {  // triangle.cpp:2068-2101
    DG_LDS DG_VOLATILE DTri &t = M.T[slot];
    t.nbr[0] = 0, t.nbr[1] = 0, t.nbr[2] = 0;
    t.vtx[0] = (uint16_t)GHOST, t.vtx[1] = (uint16_t)GHOST, t.vtx[2] = (uint16_t)GHOST;
    return slot << 2;
}

// triangle.cpp:5670-5815, the two- and three-vertex cases; a[] = vertex ids, slots [slot, slot + 2) resp. [slot, slot + 4)
__device__ __forceinline__ void d_leaf(const Mesh &M, DG_LDS const uint16_t *a, int n, uint32_t slot, uint32_t &farleft, uint32_t &farright) {
    if (n == 2) {
        uint32_t l = d_make(M, slot), r = d_make(M, slot + 1);
        D_ORG(l) = a[0];
        D_DEST(l) = a[1];
        D_ORG(r) = a[1];
        D_DEST(r) = a[0];
        D_BOND(l, r);
        l = hprev(l);
        r = hnext(r);
        D_BOND(l, r);
        l = hprev(l);
        r = hnext(r);
        D_BOND(l, r);
        farright = r;
        farleft = hprev(r);
        return;
    }
    uint32_t mid = d_make(M, slot), t1 = d_make(M, slot + 1), t2 = d_make(M, slot + 2), t3 = d_make(M, slot + 3);
    const int64_t area = d_orient(M, a[0], a[1], a[2]);
    if (area == 0) {
        D_ORG(mid) = a[0];
        D_DEST(mid) = a[1];
        D_ORG(t1) = a[1];
        D_DEST(t1) = a[0];
        D_ORG(t2) = a[2];
        D_DEST(t2) = a[1];
        D_ORG(t3) = a[1];
        D_DEST(t3) = a[2];
        D_BOND(mid, t1);
        D_BOND(t2, t3);
        mid = hnext(mid);
        t1 = hprev(t1);
        t2 = hnext(t2);
        t3 = hprev(t3);
        D_BOND(mid, t3);
        D_BOND(t1, t2);
        mid = hnext(mid);
        t1 = hprev(t1);
        t2 = hnext(t2);
        t3 = hprev(t3);
        D_BOND(mid, t1);
        D_BOND(t2, t3);
        farleft = t1;
        farright = t2;
    } else {
        const uint16_t second = area > 0 ? a[1] : a[2], third = area > 0 ? a[2] : a[1];
        D_ORG(mid) = a[0];
        D_DEST(t1) = a[0];
        D_ORG(t3) = a[0];
        D_DEST(mid) = second;
        D_ORG(t1) = second;
        D_DEST(t2) = second;
        D_APEX(mid) = third;
        D_ORG(t2) = third;
        D_DEST(t3) = third;
        D_BOND(mid, t1);
        mid = hnext(mid);
        D_BOND(mid, t2);
        mid = hnext(mid);
        D_BOND(mid, t3);
        t1 = hprev(t1);
        t2 = hnext(t2);
        D_BOND(t1, t2);
        t1 = hprev(t1);
        t3 = hprev(t3);
        D_BOND(t1, t3);
        t2 = hnext(t2);
        t3 = hprev(t3);
        D_BOND(t2, t3);
        farleft = t1;
        farright = area > 0 ? t2 : hnext(farleft);
    }
}

// triangle.cpp:5362-5651; the two new triangles take slots `slot` and `slot + 1`
__device__ __forceinline__ void d_merge(const Mesh &M, uint32_t &farleft, uint32_t innerleft, uint32_t innerright, uint32_t &farright, int axis, uint32_t slot) {
    uint32_t ild = D_DEST(innerleft), ila = D_APEX(innerleft);
    uint32_t iro = D_ORG(innerright), ira = D_APEX(innerright);
    if (axis == 1) {  // horizontal cut: walk the four extreme handles to the bottom-/top-most hull vertices
        uint32_t flp = D_ORG(farleft), fla = D_APEX(farleft);
        uint32_t frp = D_DEST(farright);
        while (D_PY(fla) < D_PY(flp)) {
            farleft = D_SYM(hnext(farleft));
            flp = fla;
            fla = D_APEX(farleft);
        }
        uint32_t chk = D_SYM(innerleft);
        uint32_t cv = D_APEX(chk);
        while (D_PY(cv) > D_PY(ild)) {
            innerleft = hnext(chk);
            ila = ild;
            ild = cv;
            chk = D_SYM(innerleft);
            cv = D_APEX(chk);
        }
        while (D_PY(ira) < D_PY(iro)) {
            innerright = D_SYM(hnext(innerright));
            iro = ira;
            ira = D_APEX(innerright);
        }
        chk = D_SYM(farright);
        cv = D_APEX(chk);
        while (D_PY(cv) > D_PY(frp)) {
            farright = hnext(chk);
            frp = cv;
            chk = D_SYM(farright);
            cv = D_APEX(chk);
        }
    }
    for (bool changed = true; changed;) {  // lower common tangent
        changed = false;
        if (d_orient(M, ild, ila, iro) > 0) {
            innerleft = D_SYM(hprev(innerleft));
            ild = ila;
            ila = D_APEX(innerleft);
            changed = true;
        }
        if (d_orient(M, ira, iro, ild) > 0) {
            innerright = D_SYM(hnext(innerright));
            iro = ira;
            ira = D_APEX(innerright);
            changed = true;
        }
    }
    uint32_t leftcand = D_SYM(innerleft), rightcand = D_SYM(innerright);
    uint32_t base = d_make(M, slot);
    D_BOND(base, innerleft);
    base = hnext(base);
    D_BOND(base, innerright);
    base = hnext(base);
    D_ORG(base) = (uint16_t)iro;
    D_DEST(base) = (uint16_t)ild;
    if (ild == D_ORG(farleft)) farleft = hnext(base);
    if (iro == D_DEST(farright)) farright = hprev(base);
    // The seam loop keeps the coordinates of its four moving vertices in registers (a vertex is read from LDS once, when it
    // enters the picture) instead of fetching them again for every predicate.
    DV ll = d_vertex(M, ild), lr = d_vertex(M, iro);
    DV ul = d_vertex(M, D_APEX(leftcand)), ur = d_vertex(M, D_APEX(rightcand));
    for (;;) {
        const bool leftdone = dv_orient(ul, ll, lr) <= 0, rightdone = dv_orient(ur, ll, lr) <= 0;
        if (leftdone && rightdone) {
            uint32_t top = d_make(M, slot + 1);
            D_ORG(top) = (uint16_t)ll.id;
            D_DEST(top) = (uint16_t)lr.id;
            D_BOND(top, base);
            top = hnext(top);
            D_BOND(top, rightcand);
            top = hnext(top);
            D_BOND(top, leftcand);
            if (axis == 1) {  // back to left-/right-most handles
                uint32_t flp = D_ORG(farleft);
                uint32_t frp = D_DEST(farright), fra = D_APEX(farright);
                uint32_t chk = D_SYM(farleft);
                uint32_t cv = D_APEX(chk);
                while (D_PX(cv) < D_PX(flp)) {
                    farleft = hprev(chk);
                    flp = cv;
                    chk = D_SYM(farleft);
                    cv = D_APEX(chk);
                }
                while (D_PX(fra) > D_PX(frp)) {
                    farright = D_SYM(hprev(farright));
                    frp = fra;
                    fra = D_APEX(farright);
                }
            }
            return;
        }
        if (!leftdone) {  // flip away left edges that the circle through ll, lr, ul invalidates
            uint32_t nx = D_SYM(hprev(leftcand));
            DV na = d_vertex(M, D_APEX(nx));
            if (na.id != GHOST) {
                bool bad = dv_incirc(ll, lr, ul, na) > 0;
                while (bad) {
                    nx = hnext(nx);
                    const uint32_t topc = D_SYM(nx);
                    nx = hnext(nx);
                    const uint32_t sidec = D_SYM(nx);
                    D_BOND(nx, topc);
                    D_BOND(leftcand, sidec);
                    leftcand = hnext(leftcand);
                    const uint32_t outerc = D_SYM(leftcand);
                    nx = hprev(nx);
                    D_BOND(nx, outerc);
                    D_ORG(leftcand) = (uint16_t)ll.id;
                    D_DEST(leftcand) = (uint16_t)GHOST;
                    D_APEX(leftcand) = (uint16_t)na.id;
                    D_ORG(nx) = (uint16_t)GHOST;
                    D_DEST(nx) = (uint16_t)ul.id;
                    D_APEX(nx) = (uint16_t)na.id;
                    ul = na;
                    nx = sidec;
                    na = d_vertex(M, D_APEX(nx));
                    bad = na.id != GHOST && dv_incirc(ll, lr, ul, na) > 0;
                }
            }
        }
        if (!rightdone) {
            uint32_t nx = D_SYM(hnext(rightcand));
            DV na = d_vertex(M, D_APEX(nx));
            if (na.id != GHOST) {
                bool bad = dv_incirc(ll, lr, ur, na) > 0;
                while (bad) {
                    nx = hprev(nx);
                    const uint32_t topc = D_SYM(nx);
                    nx = hprev(nx);
                    const uint32_t sidec = D_SYM(nx);
                    D_BOND(nx, topc);
                    D_BOND(rightcand, sidec);
                    rightcand = hprev(rightcand);
                    const uint32_t outerc = D_SYM(rightcand);
                    nx = hnext(nx);
                    D_BOND(nx, outerc);
                    D_ORG(rightcand) = (uint16_t)GHOST;
                    D_DEST(rightcand) = (uint16_t)lr.id;
                    D_APEX(rightcand) = (uint16_t)na.id;
                    D_ORG(nx) = (uint16_t)ur.id;
                    D_DEST(nx) = (uint16_t)GHOST;
                    D_APEX(nx) = (uint16_t)na.id;
                    ur = na;
                    nx = sidec;
                    na = d_vertex(M, D_APEX(nx));
                    bad = na.id != GHOST && dv_incirc(ll, lr, ur, na) > 0;
                }
            }
        }
        if (leftdone || (!rightdone && dv_incirc(ul, ll, lr, ur) > 0)) {
            D_BOND(base, rightcand);
            base = hprev(rightcand);
            D_DEST(base) = (uint16_t)ll.id;
            lr = ur;
            rightcand = D_SYM(base);
            ur = d_vertex(M, D_APEX(rightcand));
        } else {
            D_BOND(base, leftcand);
            base = hnext(leftcand);
            D_ORG(base) = (uint16_t)lr.id;
            ll = ul;
            leftcand = D_SYM(base);
            ul = d_vertex(M, D_APEX(leftcand));
        }
    }
}

// Node j of depth d of the recursion over m vertices: its vertex range [lo, lo + n), the first slot of its subtree and its
// cut axis.  False if the tree has no such node (an ancestor already is a leaf).
__device__ __forceinline__ bool d_node(int m, int d, int j, int &lo, int &n, uint32_t &slot, int &axis) {
    lo = 0;
    n = m;
    slot = 1;
    axis = 0;
    for (int b = d - 1; b >= 0; b--) {
        if (n <= 3) return false;
        const int nl = n >> 1;
        if ((j >> b) & 1) {
            slot += 2 * nl - 2;
            lo += nl;
            n -= nl;
        } else {
            n = nl;
        }
        axis ^= 1;
    }
    return true;
}

// Leaf construction or merge of node j of depth d; res[] holds (farleft | farright << 16) per node in heap order.
__device__ __forceinline__ void d_process_node(const Mesh &M, DG_LDS uint32_t *res, DG_LDS const uint16_t *ord, int m, int d, int j) {
    int lo, n, axis;
    uint32_t slot;
    if (!d_node(m, d, j, lo, n, slot, axis)) return;
    uint32_t fl, fr;
    if (n <= 3) {
        d_leaf(M, ord + lo, n, slot, fl, fr);
    } else {
        const uint32_t rl = res[(2 << d) + 2 * j], rr = res[(2 << d) + 2 * j + 1];  // children: heap index 2h, 2h+1 with h = (1<<d)+j
        fl = rl & 0xFFFFu;
        fr = rr >> 16;
        d_merge(M, fl, rl >> 16, rr & 0xFFFFu, fr, axis, slot + 2 * n - 4);
    }
    res[(1 << d) + j] = fl | (fr << 16);
}

#ifndef DG_HOST_EMULATION
// The body of both kernels: one workgroup triangulates one vertex set.
//   order[0..m)            vertex ids in k-d order (global memory)
//   vertex i               x = xb[i*stride] - (db ? db[i*stride] : 0), y = yb[i*stride], i < npts
//   out / count            triangle list (3 ids per triangle, pool order, no bounding triangles) and its length
// Depth of the deepest leaves: the smallest D with ceil(m / 2^D) <= 3.
__host__ __device__ inline int dg_depth(int m) {
    int D = 0;
    while (((m + (1 << D) - 1) >> D) > 3) D++;
    return D;
}

__device__ __forceinline__ void dg_triangulate(int m, int npts, const int32_t *__restrict__ order, const int32_t *__restrict__ xb, const int32_t *__restrict__ yb,
                                               const int32_t *__restrict__ db, int stride, int32_t *__restrict__ out, int32_t *__restrict__ count) {
    extern __shared__ uint32_t dg_lds[];
    const int tid = threadIdx.x;
    if (m < 3) {
        if (tid == 0) *count = 0;
        return;
    }
    const int nslots = 2 * m - 1;  // slot 0 + 2m - 2
    const int depth = dg_depth(m);
    // LDS: results of the tree nodes (heap order), triangles, coordinates, k-d ordered ids
    DG_LDS uint32_t *res = (DG_LDS uint32_t *)dg_lds;                       // [2 << depth]: farleft | farright << 16
    DG_LDS DG_VOLATILE DTri *T = (DG_LDS DG_VOLATILE DTri *)(res + (2 << depth));   // [nslots]
    DG_LDS int16_t *px = (DG_LDS int16_t *)(T + nslots + (nslots & 1));
    DG_LDS int16_t *py = px + npts + (npts & 1);
    DG_LDS uint16_t *ord = (DG_LDS uint16_t *)(py + npts + (npts & 1));
    for (int i = tid; i < npts; i += DG_THREADS) {
        px[i] = (int16_t)(xb[(size_t)i * stride] - (db ? db[(size_t)i * stride] : 0));
        py[i] = (int16_t)yb[(size_t)i * stride];
    }
    for (int i = tid; i < m; i += DG_THREADS) ord[i] = (uint16_t)order[i];
    if (tid == 0) {
        T[0].nbr[0] = T[0].nbr[1] = T[0].nbr[2] = 0;
        T[0].vtx[0] = T[0].vtx[1] = T[0].vtx[2] = (uint16_t)GHOST;
    }
    __syncthreads();
    const Mesh M{T, px, py};
    for (int d = depth; d >= 0; d--) {
        for (int j = tid; j < (1 << d); j += DG_THREADS) d_process_node(M, res, ord, m, d, j);
        __syncthreads();
    }
    // output in pool order without the bounding triangles: corners org / dest / apex at orientation 0
    __shared__ int s_total;
    __shared__ int s_part[DG_THREADS];
    const int per = (nslots + DG_THREADS - 1) / DG_THREADS;
    const int t0 = min(1 + tid * per, nslots), t1 = min(t0 + per, nslots);
    int mine = 0;
    for (int t = t0; t < t1; t++) mine += (T[t].vtx[0] != GHOST && T[t].vtx[1] != GHOST && T[t].vtx[2] != GHOST) ? 1 : 0;
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < DG_THREADS; i++) {
            const int c = s_part[i];
            s_part[i] = acc;
            acc += c;
        }
        s_total = acc;
    }
    __syncthreads();
    int pos = s_part[tid];
    for (int t = t0; t < t1; t++) {
        if (T[t].vtx[0] == GHOST || T[t].vtx[1] == GHOST || T[t].vtx[2] == GHOST) continue;
        out[3 * pos] = T[t].vtx[1];
        out[3 * pos + 1] = T[t].vtx[2];
        out[3 * pos + 2] = T[t].vtx[0];
        pos++;
    }
    if (tid == 0) *count = s_total;
}

// Test-hook form: sets[s] = {offset of the set's first entry in `order` / `xy`, m (vertices after the duplicate scan), n_points
// (entries of xy), offset of its triangle list in tri_out}.  order: vertex ids in k-d order; xy: (x, y) per id.
__global__ __launch_bounds__(DG_THREADS) void k_delaunay(const int4 *__restrict__ sets, const int32_t *__restrict__ order, const int32_t *__restrict__ xy,
                                                        int32_t *__restrict__ tri_out, int32_t *__restrict__ tri_count) {
    const int4 st = sets[blockIdx.x];
    dg_triangulate(st.y, st.z, order + st.x, xy + 2 * (size_t)st.x, xy + 2 * (size_t)st.x + 1, nullptr, 2, tri_out + st.w, tri_count + blockIdx.x);
}

// Pipeline form: blockIdx.x = pair * 2 + side.  The chunk's blob (engine.cpp) holds per pair the meta words, the support points
// (u, v, d) and room for the two triangle lists; behind the second list the host stage has left, per side, [m, id_0 .. id_{m-1}]:
// the k-d ordered vertex ids (m = -1: the host triangulated this side itself, nothing to do).  Left image: (u, v); right: (u - d, v).
__global__ __launch_bounds__(DG_THREADS) void k_delaunay_blob(int32_t *__restrict__ blob) {
    const int pair = blockIdx.x >> 1, side = blockIdx.x & 1;
    int32_t *meta = blob + (size_t)pair * META_WORDS;
    const int ns = meta[0];
    if (ns < 3 || meta[7] == 0) return;  // meta[7]: this pair is triangulated here (the host pool did the others)
    const int32_t *sup = blob + meta[1];
    const int32_t *ord = blob + meta[5] + (size_t)2 * ns * 3 + (size_t)side * (ns + 1);
    const int m = ord[0];
    if (m < 0) return;
    dg_triangulate(m, ns, ord + 1, sup, sup + 1, side ? sup + 2 : nullptr, 3, blob + meta[3 + 2 * side], meta + 2 + 2 * side);
}

#endif  // DG_HOST_EMULATION

}  // namespace dg

using namespace dg;

#ifndef DG_HOST_EMULATION
size_t delaunay_gpu_lds_bytes(int m, int npts) {
    const size_t nslots = 2 * (size_t)m - 1;
    return sizeof(uint32_t) * (2 << dg_depth(m)) + sizeof(DTri) * (nslots + (nslots & 1)) + sizeof(int16_t) * 2 * ((size_t)npts + (npts & 1)) + sizeof(uint16_t) * (size_t)m + 16;
}

int delaunay_gpu_max_points() { return 4000; }

// Launches one workgroup per set; lds = the largest delaunay_gpu_lds_bytes among them (<= 160 KB).
int launch_delaunay_gpu(const int4 *sets, int nsets, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024) {
        static std::atomic<size_t> granted[64];  // per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t> &g = granted[dev & 63];
        if (lds > g.load()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_delaunay), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
            g.store(lds);
        }
    }
    hipLaunchKernelGGL(k_delaunay, dim3(nsets), dim3(DG_THREADS), lds, st, sets, order, xy, tri_out, tri_count);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Both triangulations of every pair of a chunk, straight into the chunk's device blob.
void launch_delaunay_blob(int32_t *blob, int n_pairs, size_t lds, hipStream_t st) {
    static std::atomic<size_t> granted[64];
    ensure_dynamic_lds(k_delaunay_blob, lds, granted, "delaunay_gpu");
    SV_LAUNCH(K_DELAUNAY, k_delaunay_blob, dim3(2 * n_pairs), dim3(DG_THREADS), lds, st, blob);
}
#endif  // DG_HOST_EMULATION

}  // namespace sv
