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
// Sets too large for LDS (> DG_SUB_MAX vertices, the 4K configuration: ~30 000) are cut at the tree depth c where every node fits:
// k_dgl_subtrees triangulates the 2^c subtrees as above, one workgroup each with local vertex numbers, and exports them into a
// mesh in global memory (24-byte triangles with 32-bit handles, slot numbers of the whole set); k_dgl_top then runs the 2^c - 1
// remaining merges on that mesh, depth by depth, one lane per merge (the seams are O(sqrt n) long), and compacts the list.
//
// Input: the vertices in the k-d order the reference's alternating cuts leave (host: radix sort, duplicate scan, kd_order).
// Output: the triangle list in pool order without the bounding ("ghost") triangles (triangle.cpp:7449-7500).
#ifndef DG_HOST_EMULATION  // tests/emu_delaunay_gpu.cpp compiles the device functions for the CPU
#include <hip/hip_runtime.h>

#include <atomic>

#include "sv_kernels.h"
#endif
#include <stddef.h>
#include <stdint.h>

// The workgroup-cooperative code (the on-GPU preparation of a vertex set) is compiled for the device and - DG_SIMT_EMULATION, on top of
// DG_HOST_EMULATION - for CPU threads that play one workgroup under the sanitizers: tests/emu_dg_prepare.cpp supplies threadIdx,
// __syncthreads, __shfl_up and the atomics.
#if !defined(DG_HOST_EMULATION) || defined(DG_SIMT_EMULATION)
#define DG_COOP 1
#endif

namespace sv {

namespace dg {

constexpr uint32_t GHOST = 0xFFFFu;
constexpr int DG_THREADS = 256;

// A triangle of an LDS mesh: three neighbour handles ((slot << 2) | orientation, slot 0 = outer space) and three vertex ids (GHOST = the
// vertex at infinity of a bounding triangle), 16 bits each, as three 32-bit words:
//   w[0] = nbr0 | nbr1 << 16    w[1] = nbr2 | vtx0 << 16    w[2] = vtx1 | vtx2 << 16
// A merge is one lane's chain of dependent LDS accesses (~30 ns each, tools/valu_rate2.hip), so the seam walk reads a triangle as ONE
// access - three ds_read_b32 in flight together - into registers (TR) and picks its fields there; round 3 read every 16-bit field by
// itself (volatile, a full wait each): ~3 us per seam step against ~0.4 us now.  Stores stay single 16-bit fields.
struct DTri {
    uint32_t w[3];
};
struct TR {  // a triangle in registers
    uint32_t w0, w1, w2;
};
__device__ __forceinline__ uint32_t tr_nbr(const TR &t, uint32_t o) { return (uint32_t)(((((uint64_t)t.w1) << 32) | t.w0) >> (16u * o)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t tr_vtx(const TR &t, uint32_t k) { return (uint32_t)(((((uint64_t)t.w2) << 32) | t.w1) >> (16u * k + 16u)) & 0xFFFFu; }

#ifdef DG_HOST_EMULATION
#define DG_LDS
#else
#define DG_LDS __attribute__((address_space(3)))  // explicit LDS pointers: ds_* instructions instead of flat_* ones
#endif
// The loads are inline assembly on the device: hipcc's SLP vectoriser (ROCm 7.2, -O2 and above) fuses plain field accesses of
// neighbouring statements into wider ones ACROSS stores that may hit the same triangle through another handle (reproduced on a
// 7-vertex set in round 2), and a volatile access is followed by a full wait.  LDS operations of one wavefront execute in issue order,
// so a load issued after a store sees it; "memory" keeps the compiler from moving either across the other.
struct Mesh {  // a set (or a subtree of a large set) in LDS
    DG_LDS uint32_t *W;          // triangles, 3 words each
    DG_LDS const uint32_t *pxy;  // vertices: (x & 0xFFFF) | y << 16, both signed 16-bit
#ifdef DG_HOST_EMULATION
    TR load(uint32_t slot) const { return TR{W[3 * slot], W[3 * slot + 1], W[3 * slot + 2]}; }
    void load2(uint32_t a, uint32_t b, TR &ta, TR &tb) const { ta = load(a), tb = load(b); }
    uint32_t coords(uint32_t v) const { return pxy[v]; }
    void coords2(uint32_t a, uint32_t b, uint32_t &ca, uint32_t &cb) const { ca = pxy[a], cb = pxy[b]; }
    uint32_t field(uint32_t slot, uint32_t f) const { return reinterpret_cast<const uint16_t *>(W)[6 * slot + f]; }
    void set_field(uint32_t slot, uint32_t f, uint32_t v) const { reinterpret_cast<uint16_t *>(W)[6 * slot + f] = (uint16_t)v; }
    void store(uint32_t slot, const TR &t) const { W[3 * slot] = t.w0, W[3 * slot + 1] = t.w1, W[3 * slot + 2] = t.w2; }
#else
    __device__ __forceinline__ TR load(uint32_t slot) const {
        TR t;
        const uint32_t a = (uint32_t)(size_t)W + 12u * slot;
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:4\n\tds_read_b32 %2, %3 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t.w0), "=&v"(t.w1), "=&v"(t.w2) : "v"(a) : "memory");
        return t;
    }
    __device__ __forceinline__ void load2(uint32_t sa, uint32_t sb, TR &ta, TR &tb) const {  // two independent triangles, one latency
        const uint32_t a = (uint32_t)(size_t)W + 12u * sa, b = (uint32_t)(size_t)W + 12u * sb;
        asm volatile(
            "ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:4\n\tds_read_b32 %2, %6 offset:8\n\tds_read_b32 %3, %7\n\tds_read_b32 %4, %7 offset:4\n\tds_read_b32 %5, %7 offset:8\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(ta.w0), "=&v"(ta.w1), "=&v"(ta.w2), "=&v"(tb.w0), "=&v"(tb.w1), "=&v"(tb.w2)
            : "v"(a), "v"(b)
            : "memory");
    }
    __device__ __forceinline__ uint32_t coords(uint32_t v) const {
        uint32_t c;
        const uint32_t a = (uint32_t)(size_t)pxy + 4u * v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(a) : "memory");
        return c;
    }
    __device__ __forceinline__ void coords2(uint32_t va, uint32_t vb, uint32_t &ca, uint32_t &cb) const {
        const uint32_t a = (uint32_t)(size_t)pxy + 4u * va, b = (uint32_t)(size_t)pxy + 4u * vb;
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ca), "=&v"(cb) : "v"(a), "v"(b) : "memory");
    }
    __device__ __forceinline__ uint32_t field(uint32_t slot, uint32_t f) const {  // one 16-bit field (f = 0..2: neighbours, 3..5: vertices)
        uint32_t c;
        const uint32_t a = (uint32_t)(size_t)W + 12u * slot + 2u * f;
        asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(a) : "memory");
        return c;
    }
    __device__ __forceinline__ void set_field(uint32_t slot, uint32_t f, uint32_t v) const {
        const uint32_t a = (uint32_t)(size_t)W + 12u * slot + 2u * f;
        asm volatile("ds_write_b16 %0, %1" : : "v"(a), "v"(v) : "memory");
    }
    __device__ __forceinline__ void store(uint32_t slot, const TR &t) const {
        const uint32_t a = (uint32_t)(size_t)W + 12u * slot;
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8" : : "v"(a), "v"(t.w0), "v"(t.w1), "v"(t.w2) : "memory");
    }
#endif
    __device__ __forceinline__ void bond(uint32_t a, uint32_t b) const {  // triangle.cpp: bond(): the two handles become each other's neighbour
        set_field(a >> 2, a & 3u, b);
        set_field(b >> 2, b & 3u, a);
    }
    // the interface d_merge_mesh is written against (MeshG below has the same)
    typedef TR TRT;
    static constexpr uint32_t NOV = GHOST;      // "no vertex": the vertex at infinity
    static constexpr bool GUARDED = false;      // (a merge on a mesh in LDS sees exactly what this workgroup wrote)
    static constexpr uint32_t step_limit = 0;
    static __device__ __forceinline__ uint32_t nbr(const TR &t, uint32_t o) { return tr_nbr(t, o); }
    static __device__ __forceinline__ uint32_t vtx(const TR &t, uint32_t k) { return tr_vtx(t, k); }
    static __device__ __forceinline__ TR pack(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t v0, uint32_t v1, uint32_t v2) {
        return TR{(n0 & 0xFFFFu) | (n1 << 16), (n2 & 0xFFFFu) | (v0 << 16), (v1 & 0xFFFFu) | (v2 << 16)};
    }
};

// The same triangle in a global-memory mesh (the upper levels of a set that does not fit LDS): six 32-bit words - three neighbour
// handles, three vertex ids of the set (GHOST32 = the vertex at infinity).  The merges of a set run in ONE workgroup (what the subtree
// kernel wrote is visible since the kernel boundary), so its accesses are relaxed WORKGROUP-scope atomics: single 32-bit accesses that
// keep their order per address, are left alone by the SLP vectoriser (see Mesh), overlap when they are independent, and may use the CU's
// vector cache.  (Round 3 had them at agent scope - every field a round trip to L2 - and read field by field.)
#define DG_VOLATILE volatile
struct GTri {
    uint32_t w[6];  // nbr0 nbr1 nbr2 vtx0 vtx1 vtx2
};
constexpr uint32_t GHOST32 = 0xFFFFFFFFu;
struct TRG {  // ... in registers
    uint32_t n0, n1, n2, v0, v1, v2;
};

struct MeshG {
    GTri *T;
    const int32_t *xy;  // (x, y) per vertex id
    // A merge on this mesh reads what other workgroups / an earlier kernel wrote; should it ever see a torn mesh, its walks may
    // not end.  Every loop trip of a merge counts against step_limit (far above any real merge): a wrong list, never a hung GPU.
    uint32_t step_limit;
    typedef TRG TRT;
    static constexpr uint32_t NOV = GHOST32;
    static constexpr bool GUARDED = true;
#ifdef DG_HOST_EMULATION
    static uint32_t ldw(const uint32_t *p) { return *p; }
    static void stw(uint32_t *p, uint32_t v) { *p = v; }
#else
    static __device__ __forceinline__ uint32_t ldw(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    static __device__ __forceinline__ void stw(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
    __device__ __forceinline__ TRG load(uint32_t slot) const {
        const uint32_t *w = T[slot].w;
        return TRG{ldw(w), ldw(w + 1), ldw(w + 2), ldw(w + 3), ldw(w + 4), ldw(w + 5)};
    }
    __device__ __forceinline__ void load2(uint32_t a, uint32_t b, TRG &ta, TRG &tb) const { ta = load(a), tb = load(b); }
    __device__ __forceinline__ uint32_t field(uint32_t slot, uint32_t f) const { return ldw(T[slot].w + f); }
    __device__ __forceinline__ void set_field(uint32_t slot, uint32_t f, uint32_t v) const { stw(T[slot].w + f, v); }
    __device__ __forceinline__ void store(uint32_t slot, const TRG &t) const {
        uint32_t *w = T[slot].w;
        stw(w, t.n0), stw(w + 1, t.n1), stw(w + 2, t.n2), stw(w + 3, t.v0), stw(w + 4, t.v1), stw(w + 5, t.v2);
    }
    __device__ __forceinline__ void bond(uint32_t a, uint32_t b) const {
        set_field(a >> 2, a & 3u, b);
        set_field(b >> 2, b & 3u, a);
    }
    static __device__ __forceinline__ uint32_t nbr(const TRG &t, uint32_t o) { return o == 0 ? t.n0 : (o == 1 ? t.n1 : t.n2); }
    static __device__ __forceinline__ uint32_t vtx(const TRG &t, uint32_t k) { return k == 0 ? t.v0 : (k == 1 ? t.v1 : t.v2); }
    static __device__ __forceinline__ TRG pack(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t v0, uint32_t v1, uint32_t v2) { return TRG{n0, n1, n2, v0, v1, v2}; }
};

__device__ __forceinline__ uint32_t next3(uint32_t o) { return (9u >> (2 * o)) & 3u; }
__device__ __forceinline__ uint32_t prev3(uint32_t o) { return (18u >> (2 * o)) & 3u; }
__device__ __forceinline__ uint32_t hnext(uint32_t h) { return (h & ~3u) | next3(h & 3u); }
__device__ __forceinline__ uint32_t hprev(uint32_t h) { return (h & ~3u) | prev3(h & 3u); }

// ---- LDS mesh: vertices with coordinates, predicates ------------------------------------------------------------------------
struct DVL {  // a vertex of an LDS mesh with its coordinates (the GHOST vertex has none: they are never used)
    uint32_t id;
    int32_t x, y;
};
__device__ __forceinline__ DVL dvl(uint32_t id, uint32_t c) { return DVL{id, (int32_t)(int16_t)(c & 0xFFFFu), (int32_t)c >> 16}; }
__device__ __forceinline__ uint32_t vsafe(uint32_t id) { return id == GHOST ? 0u : id; }
__device__ __forceinline__ DVL ld_vertex(const Mesh &M, uint32_t id) { return dvl(id, M.coords(vsafe(id))); }
__device__ __forceinline__ void ld_vertex2(const Mesh &M, uint32_t ia, uint32_t ib, DVL &a, DVL &b) {
    uint32_t ca, cb;
    M.coords2(vsafe(ia), vsafe(ib), ca, cb);
    a = dvl(ia, ca), b = dvl(ib, cb);
}
__device__ __forceinline__ DVL ld_vertex(const MeshG &M, uint32_t id) {  // (coordinates are read-only here: plain loads)
    const size_t v = id == GHOST32 ? 0u : id;
    return DVL{id, M.xy[2 * v], M.xy[2 * v + 1]};
}
__device__ __forceinline__ void ld_vertex2(const MeshG &M, uint32_t ia, uint32_t ib, DVL &a, DVL &b) { a = ld_vertex(M, ia), b = ld_vertex(M, ib); }

// Orientation: coordinates are 16-bit, the two products 32 x 32 -> 64 bits (v_mad_i64_i32).
__device__ __forceinline__ int64_t lv_orient(const DVL &a, const DVL &b, const DVL &c) {
    return (int64_t)(a.x - c.x) * (int64_t)(b.y - c.y) - (int64_t)(a.y - c.y) * (int64_t)(b.x - c.x);
}
// In-circle.  NARROW: every coordinate difference of the set is below 2^14 in magnitude (the engine's sets: columns within
// [-disp_max, W + disp_max), W <= 8192, disp_max <= 1023) - then the lifted terms (< 2^29) and the 2 x 2 minors (< 2^29) are exact in
// 32 bits and only the three final products need 64: ~25 instructions instead of ~150.
template <bool NARROW>
__device__ __forceinline__ int64_t lv_incirc(const DVL &a, const DVL &b, const DVL &c, const DVL &d) {
    if (NARROW) {
        const int32_t adx = a.x - d.x, ady = a.y - d.y, bdx = b.x - d.x, bdy = b.y - d.y, cdx = c.x - d.x, cdy = c.y - d.y;
        const int32_t al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
        return (int64_t)al * (int64_t)(bdx * cdy - cdx * bdy) + (int64_t)bl * (int64_t)(cdx * ady - adx * cdy) + (int64_t)cl * (int64_t)(adx * bdy - bdx * ady);
    }
    const int64_t adx = a.x - d.x, ady = a.y - d.y, bdx = b.x - d.x, bdy = b.y - d.y, cdx = c.x - d.x, cdy = c.y - d.y;
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
}

// A triangle assembled in registers from its six fields (the leaves and the two new triangles of a merge are stored whole)
__device__ __forceinline__ TR tr_pack(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t v0, uint32_t v1, uint32_t v2) { return Mesh::pack(n0, n1, n2, v0, v1, v2); }

// triangle.cpp:5670-5815, the two- and three-vertex cases; a[] = vertex ids, slots [slot, slot + 2) resp. [slot, slot + 4).
// The bonds and corners below are what that code leaves (restated field by field from the round-3 kernel, which followed it handle
// by handle); H(s, o) = handle of slot s at orientation o.
__device__ __forceinline__ void d_leaf(const Mesh &M, DG_LDS const uint16_t *a, int n, uint32_t slot, uint32_t &farleft, uint32_t &farright) {
#define DG_H(s, o) ((((uint32_t)(s)) << 2) | (uint32_t)(o))
    const uint32_t a0 = a[0], a1 = a[1];
    if (n == 2) {
        // l = slot: org a0 (vtx[1]), dest a1 (vtx[2]); r = slot + 1: org a1, dest a0.  Bonds: (l,0)-(r,0), (l,2)-(r,1), (l,1)-(r,2).
        const uint32_t l = slot, r = slot + 1;
        M.store(l, tr_pack(DG_H(r, 0), DG_H(r, 2), DG_H(r, 1), GHOST, a0, a1));
        M.store(r, tr_pack(DG_H(l, 0), DG_H(l, 2), DG_H(l, 1), GHOST, a1, a0));
        farright = DG_H(r, 2);  // r after two hnext: 0 -> 1 -> 2
        farleft = DG_H(r, 1);   // hprev of that
        return;
    }
    const uint32_t a2 = a[2];
    DVL v0, v1;
    ld_vertex2(M, a0, a1, v0, v1);
    const DVL v2 = ld_vertex(M, a2);
    const int64_t area = lv_orient(v0, v1, v2);
    const uint32_t mid = slot, t1 = slot + 1, t2 = slot + 2, t3 = slot + 3;
    if (area == 0) {
        // three collinear vertices: two edges, four bounding triangles.  Corners: mid org a0 dest a1; t1 org a1 dest a0; t2 org a2 dest a1;
        // t3 org a1 dest a2.  Bonds: (mid,0)-(t1,0), (t2,0)-(t3,0), (mid,1)-(t3,2), (t1,2)-(t2,1), (mid,2)-(t1,1), (t2,2)-(t3,1).
        M.store(mid, tr_pack(DG_H(t1, 0), DG_H(t3, 2), DG_H(t1, 1), GHOST, a0, a1));
        M.store(t1, tr_pack(DG_H(mid, 0), DG_H(mid, 2), DG_H(t2, 1), GHOST, a1, a0));
        M.store(t2, tr_pack(DG_H(t3, 0), DG_H(t1, 2), DG_H(t3, 1), GHOST, a2, a1));
        M.store(t3, tr_pack(DG_H(t2, 0), DG_H(t2, 2), DG_H(mid, 1), GHOST, a1, a2));
        farleft = DG_H(t1, 1);   // t1 after two hprev: 0 -> 2 -> 1
        farright = DG_H(t2, 2);  // t2 after two hnext: 0 -> 1 -> 2
    } else {
        // one real triangle `mid` (a0, second, third counter-clockwise) and three bounding ones.
        const uint32_t second = area > 0 ? a1 : a2, third = area > 0 ? a2 : a1;
        // mid: org a0 (vtx[1]), dest second (vtx[2]), apex third (vtx[0]); t1: org second, dest a0; t2: org third, dest second; t3: org a0, dest third
        // Bonds: (mid,0)-(t1,0), (mid,1)-(t2,0), (mid,2)-(t3,0), (t1,2)-(t2,1), (t1,1)-(t3,2), (t2,2)-(t3,1).
        M.store(mid, tr_pack(DG_H(t1, 0), DG_H(t2, 0), DG_H(t3, 0), third, a0, second));
        M.store(t1, tr_pack(DG_H(mid, 0), DG_H(t3, 2), DG_H(t2, 1), GHOST, second, a0));
        M.store(t2, tr_pack(DG_H(mid, 1), DG_H(t1, 2), DG_H(t3, 1), GHOST, third, second));
        M.store(t3, tr_pack(DG_H(mid, 2), DG_H(t2, 2), DG_H(t1, 1), GHOST, a0, third));
        farleft = DG_H(t1, 1);  // t1 after two hprev
        farright = area > 0 ? DG_H(t2, 2) : DG_H(t1, 2);  // t2 after two hnext; else hnext(farleft)
    }
#undef DG_H
}

// Node j of depth d of the recursion over m vertices: its vertex range [lo, lo + n), the first slot of its subtree and its
// cut axis (axis0 = the axis of the tree's root: 0 for a whole set).  False if the tree has no such node (an ancestor already is a leaf).
__device__ __forceinline__ bool d_node(int m, int d, int j, int &lo, int &n, uint32_t &slot, int &axis, int axis0 = 0) {
    lo = 0;
    n = m;
    slot = 1;
    axis = axis0;
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

// triangle.cpp:5362-5651 (mergehulls) on a mesh MT - Mesh (LDS, 16-bit handles) or MeshG (global memory, 32-bit handles); the two new
// triangles take slots `slot` and `slot + 1`.  Every triangle the merge looks at is held in registers (MT::TRT) from ONE access, the
// coordinates of the moving vertices are kept beside their ids, the two sides' next candidates are fetched together, and the in-circle
// test runs in 32-bit terms.  A register copy is only used until the next store that could touch its triangle; after a flip both
// triangles are read again.  MT::GUARDED: every loop trip counts against M.step_limit (a torn mesh ends the merge, never hangs the GPU).
template <bool NARROW, class MT>
__device__ __forceinline__ void d_merge_mesh(const MT &M, uint32_t &farleft, uint32_t innerleft, uint32_t innerright, uint32_t &farright, int axis, uint32_t slot) {
    typedef typename MT::TRT TR;
    constexpr uint32_t GHOST = MT::NOV;  // (shadows the 16-bit constant of the namespace)
    uint32_t steps = 0;
#define D_STEP()                                           \
    do {                                                   \
        if (MT::GUARDED && ++steps > M.step_limit) return; \
    } while (0)
    TR TL, TRr;  // the triangles of innerleft / innerright
    M.load2(innerleft >> 2, innerright >> 2, TL, TRr);
    DVL ild, ila, iro, ira;  // dest / apex of innerleft, org / apex of innerright
    ld_vertex2(M, MT::vtx(TL, prev3(innerleft & 3u)), MT::vtx(TL, innerleft & 3u), ild, ila);
    ld_vertex2(M, MT::vtx(TRr, next3(innerright & 3u)), MT::vtx(TRr, innerright & 3u), iro, ira);
    if (axis == 1) {  // horizontal cut: walk the four extreme handles to the bottom-/top-most hull vertices
        {
            TR t = M.load(farleft >> 2);
            DVL flp, fla;
            ld_vertex2(M, MT::vtx(t, next3(farleft & 3u)), MT::vtx(t, farleft & 3u), flp, fla);
            while (fla.y < flp.y) {
                D_STEP();
                farleft = MT::nbr(t, next3(farleft & 3u));
                t = M.load(farleft >> 2);
                flp = fla;
                fla = ld_vertex(M, MT::vtx(t, farleft & 3u));
            }
        }
        {
            uint32_t chk = MT::nbr(TL, innerleft & 3u);
            TR t = M.load(chk >> 2);
            DVL cv = ld_vertex(M, MT::vtx(t, chk & 3u));
            while (cv.y > ild.y) {
                D_STEP();
                innerleft = hnext(chk);
                TL = t;  // same triangle, next orientation
                ila = ild;
                ild = cv;
                chk = MT::nbr(TL, innerleft & 3u);
                t = M.load(chk >> 2);
                cv = ld_vertex(M, MT::vtx(t, chk & 3u));
            }
        }
        while (ira.y < iro.y) {
            D_STEP();
            innerright = MT::nbr(TRr, next3(innerright & 3u));
            TRr = M.load(innerright >> 2);
            iro = ira;
            ira = ld_vertex(M, MT::vtx(TRr, innerright & 3u));
        }
        {
            TR t = M.load(farright >> 2);
            DVL frp = ld_vertex(M, MT::vtx(t, prev3(farright & 3u)));
            uint32_t chk = MT::nbr(t, farright & 3u);
            t = M.load(chk >> 2);
            DVL cv = ld_vertex(M, MT::vtx(t, chk & 3u));
            while (cv.y > frp.y) {
                D_STEP();
                farright = hnext(chk);
                frp = cv;
                chk = MT::nbr(t, farright & 3u);
                t = M.load(chk >> 2);
                cv = ld_vertex(M, MT::vtx(t, chk & 3u));
            }
        }
    }
    for (bool changed = true; changed;) {  // lower common tangent
        D_STEP();
        changed = false;
        if (lv_orient(ild, ila, iro) > 0) {
            innerleft = MT::nbr(TL, prev3(innerleft & 3u));
            TL = M.load(innerleft >> 2);
            ild = ila;
            ila = ld_vertex(M, MT::vtx(TL, innerleft & 3u));
            changed = true;
        }
        if (lv_orient(ira, iro, ild) > 0) {
            innerright = MT::nbr(TRr, next3(innerright & 3u));
            TRr = M.load(innerright >> 2);
            iro = ira;
            ira = ld_vertex(M, MT::vtx(TRr, innerright & 3u));
            changed = true;
        }
    }
    uint32_t leftcand = MT::nbr(TL, innerleft & 3u), rightcand = MT::nbr(TRr, innerright & 3u);
    // the first new triangle: base = slot at orientation 2 after its two bonds; org(base) = iro (vtx[0]), dest(base) = ild (vtx[1])
    M.store(slot, MT::pack(innerleft, innerright, 0u, iro.id, ild.id, GHOST));
    M.set_field(innerleft >> 2, innerleft & 3u, (slot << 2) | 0u);
    M.set_field(innerright >> 2, innerright & 3u, (slot << 2) | 1u);
    uint32_t base = (slot << 2) | 2u;
    if (ild.id == M.field(farleft >> 2, 3u + next3(farleft & 3u))) farleft = hnext(base);
    if (iro.id == M.field(farright >> 2, 3u + prev3(farright & 3u))) farright = hprev(base);
    DVL ll = ild, lr = iro, ul, ur;
    TR TLc, TRc;  // the triangles of leftcand / rightcand
    M.load2(leftcand >> 2, rightcand >> 2, TLc, TRc);
    ld_vertex2(M, MT::vtx(TLc, leftcand & 3u), MT::vtx(TRc, rightcand & 3u), ul, ur);
    for (;;) {
        D_STEP();
        const bool leftdone = lv_orient(ul, ll, lr) <= 0, rightdone = lv_orient(ur, ll, lr) <= 0;
        if (leftdone && rightdone) {
            // the second new triangle `top`: org ll (vtx[1]), dest lr (vtx[2]); bonds (top,0)-base, (top,1)-rightcand, (top,2)-leftcand
            const uint32_t top = slot + 1;
            M.store(top, MT::pack(base, rightcand, leftcand, GHOST, ll.id, lr.id));
            M.set_field(base >> 2, base & 3u, (top << 2) | 0u);
            M.set_field(rightcand >> 2, rightcand & 3u, (top << 2) | 1u);
            M.set_field(leftcand >> 2, leftcand & 3u, (top << 2) | 2u);
            if (axis == 1) {  // back to left-/right-most handles
                {
                    TR t = M.load(farleft >> 2);
                    DVL flp = ld_vertex(M, MT::vtx(t, next3(farleft & 3u)));
                    uint32_t chk = MT::nbr(t, farleft & 3u);
                    t = M.load(chk >> 2);
                    DVL cv = ld_vertex(M, MT::vtx(t, chk & 3u));
                    while (cv.x < flp.x) {
                        D_STEP();
                        farleft = hprev(chk);
                        flp = cv;
                        chk = MT::nbr(t, farleft & 3u);
                        t = M.load(chk >> 2);
                        cv = ld_vertex(M, MT::vtx(t, chk & 3u));
                    }
                }
                {
                    TR t = M.load(farright >> 2);
                    DVL frp, fra;
                    ld_vertex2(M, MT::vtx(t, prev3(farright & 3u)), MT::vtx(t, farright & 3u), frp, fra);
                    while (fra.x > frp.x) {
                        D_STEP();
                        farright = MT::nbr(t, prev3(farright & 3u));
                        t = M.load(farright >> 2);
                        frp = fra;
                        fra = ld_vertex(M, MT::vtx(t, farright & 3u));
                    }
                }
            }
            return;
        }
        // both sides' next candidates: the triangles across the candidates' far edges and their apexes (one latency each for the pair)
        uint32_t nxl = MT::nbr(TLc, prev3(leftcand & 3u)), nxr = MT::nbr(TRc, next3(rightcand & 3u));
        TR TNl, TNr;
        M.load2(nxl >> 2, nxr >> 2, TNl, TNr);
        DVL nal, nar;
        ld_vertex2(M, MT::vtx(TNl, nxl & 3u), MT::vtx(TNr, nxr & 3u), nal, nar);
        if (!leftdone && nal.id != GHOST) {  // flip away left edges that the circle through ll, lr, ul invalidates
            bool bad = lv_incirc<NARROW>(ll, lr, ul, nal) > 0, flipped = bad;
            while (bad) {
                D_STEP();
                const uint32_t nx1 = hnext(nxl), nx2 = hnext(nx1);
                const uint32_t topc = MT::nbr(TNl, nx1 & 3u), sidec = MT::nbr(TNl, nx2 & 3u);
                M.bond(nx2, topc);
                M.bond(leftcand, sidec);
                leftcand = hnext(leftcand);
                const uint32_t outerc = M.field(leftcand >> 2, leftcand & 3u);  // (read after the stores: they may have written it)
                M.bond(nx1, outerc);
                M.set_field(leftcand >> 2, 3u + next3(leftcand & 3u), ll.id);
                M.set_field(leftcand >> 2, 3u + prev3(leftcand & 3u), GHOST);
                M.set_field(leftcand >> 2, 3u + (leftcand & 3u), nal.id);
                M.set_field(nx1 >> 2, 3u + next3(nx1 & 3u), GHOST);
                M.set_field(nx1 >> 2, 3u + prev3(nx1 & 3u), ul.id);
                M.set_field(nx1 >> 2, 3u + (nx1 & 3u), nal.id);
                ul = nal;
                nxl = sidec;
                TNl = M.load(nxl >> 2);
                nal = ld_vertex(M, MT::vtx(TNl, nxl & 3u));
                bad = nal.id != GHOST && lv_incirc<NARROW>(ll, lr, ul, nal) > 0;
            }
            if (flipped) TLc = M.load(leftcand >> 2);
        }
        if (!rightdone && nar.id != GHOST) {
            bool bad = lv_incirc<NARROW>(ll, lr, ur, nar) > 0, flipped = bad;
            while (bad) {
                D_STEP();
                const uint32_t nx1 = hprev(nxr), nx2 = hprev(nx1);
                const uint32_t topc = MT::nbr(TNr, nx1 & 3u), sidec = MT::nbr(TNr, nx2 & 3u);
                M.bond(nx2, topc);
                M.bond(rightcand, sidec);
                rightcand = hprev(rightcand);
                const uint32_t outerc = M.field(rightcand >> 2, rightcand & 3u);
                M.bond(nx1, outerc);
                M.set_field(rightcand >> 2, 3u + next3(rightcand & 3u), GHOST);
                M.set_field(rightcand >> 2, 3u + prev3(rightcand & 3u), lr.id);
                M.set_field(rightcand >> 2, 3u + (rightcand & 3u), nar.id);
                M.set_field(nx1 >> 2, 3u + next3(nx1 & 3u), ur.id);
                M.set_field(nx1 >> 2, 3u + prev3(nx1 & 3u), GHOST);
                M.set_field(nx1 >> 2, 3u + (nx1 & 3u), nar.id);
                ur = nar;
                nxr = sidec;
                TNr = M.load(nxr >> 2);
                nar = ld_vertex(M, MT::vtx(TNr, nxr & 3u));
                bad = nar.id != GHOST && lv_incirc<NARROW>(ll, lr, ur, nar) > 0;
            }
            if (flipped) TRc = M.load(rightcand >> 2);
        }
        if (leftdone || (!rightdone && lv_incirc<NARROW>(ul, ll, lr, ur) > 0)) {
            M.bond(base, rightcand);
            base = hprev(rightcand);
            M.set_field(base >> 2, 3u + prev3(base & 3u), ll.id);  // dest(base)
            lr = ur;
            rightcand = MT::nbr(TRc, base & 3u);  // sym(base): a field of the old candidate's triangle that the bond above did not write
            TRc = M.load(rightcand >> 2);
            ur = ld_vertex(M, MT::vtx(TRc, rightcand & 3u));
        } else {
            M.bond(base, leftcand);
            base = hnext(leftcand);
            M.set_field(base >> 2, 3u + next3(base & 3u), lr.id);  // org(base)
            ll = ul;
            leftcand = MT::nbr(TLc, base & 3u);
            TLc = M.load(leftcand >> 2);
            ul = ld_vertex(M, MT::vtx(TLc, leftcand & 3u));
        }
    }
}
#undef D_STEP

// Leaf construction or merge of node j of depth d; res[] holds (farleft | farright << 16) per node in heap order.
template <bool NARROW>
__device__ __forceinline__ void d_process_node(const Mesh &M, DG_LDS uint32_t *res, DG_LDS const uint16_t *ord, int m, int d, int j, int axis0 = 0) {
    int lo, n, axis;
    uint32_t slot;
    if (!d_node(m, d, j, lo, n, slot, axis, axis0)) return;
    uint32_t fl, fr;
    if (n <= 3) {
        d_leaf(M, ord + lo, n, slot, fl, fr);
    } else {
        const uint32_t rl = res[(2 << d) + 2 * j], rr = res[(2 << d) + 2 * j + 1];  // children: heap index 2h, 2h+1 with h = (1<<d)+j
        fl = rl & 0xFFFFu;
        fr = rr >> 16;
        d_merge_mesh<NARROW>(M, fl, rl >> 16, rr & 0xFFFFu, fr, axis, slot + 2 * n - 4);
    }
    res[(1 << d) + j] = fl | (fr << 16);
}

// Depth of the deepest leaves: the smallest D with ceil(m / 2^D) <= 3.
__host__ __device__ inline int dg_depth(int m) {
    int D = 0;
    while (((m + (1 << D) - 1) >> D) > 3) D++;
    return D;
}

// ---- sets that do not fit LDS ---------------------------------------------------------------------------------------------
constexpr int DG_SUB_MAX = 4000;  // vertices of a subtree (and of a whole set) triangulated in LDS
constexpr int DG_CUT_MAX = 6;     // at most 2^6 subtrees: sets of up to 256 000 vertices

// Depth at which a set of m vertices is cut: the smallest c whose largest node, ceil(m / 2^c) vertices, fits.
__host__ __device__ inline int dg_cut_depth(int m, int sub_max) {
    int c = 0;
    while (((m + (1 << c) - 1) >> c) > sub_max) c++;
    return c;
}

// A subtree's handle (local slot numbers from 1) in the numbering of the whole set (the subtree's slots start at slot0)
__device__ __forceinline__ uint32_t dg_global_handle(uint32_t h, uint32_t slot0) { return (h >> 2) == 0 ? 0u : (((h >> 2) + slot0 - 1u) << 2) | (h & 3u); }

// One of the remaining merges: node j of depth d (< cut depth) of a set of m vertices; gres[2h], gres[2h + 1] = farleft / farright of heap node h
template <bool NARROW>
__device__ __forceinline__ void dg_top_node(const MeshG &M, DG_VOLATILE uint32_t *gres, int m, int d, int j) {
    int lo, n, axis;
    uint32_t slot;
    if (!d_node(m, d, j, lo, n, slot, axis)) return;
    const int hl = (2 << d) + 2 * j, hr = hl + 1;  // children of heap node (1 << d) + j
    uint32_t fl = gres[2 * hl], fr = gres[2 * hr + 1];
    d_merge_mesh<NARROW>(M, fl, gres[2 * hl + 1], gres[2 * hr], fr, axis, slot + 2 * n - 4);
    gres[2 * ((1 << d) + j)] = fl;
    gres[2 * ((1 << d) + j) + 1] = fr;
}

#ifndef DG_HOST_EMULATION
// The body of both kernels: one workgroup triangulates one vertex set.
//   order[0..m)            vertex ids in k-d order (global memory)
//   vertex i               x = xb[i*stride] - (db ? db[i*stride] : 0), y = yb[i*stride], i < npts
//   out / count            triangle list (3 ids per triangle, pool order, no bounding triangles) and its length

// One vertex set as the kernels see it
struct DgSet {
    int m, npts;                   // vertices after the duplicate scan / entries of the coordinate arrays
    const int32_t *order;          // [m] vertex ids in k-d order
    const int32_t *xb, *yb, *db;   // vertex i: x = xb[i * stride] - (db ? db[i * stride] : 0), y = yb[i * stride]
    int stride;
    int32_t *out, *count;          // triangle list (3 ids per triangle, pool order, no bounding triangles) and its length
    __device__ __forceinline__ int x(int i) const { return xb[(size_t)i * stride] - (db ? db[(size_t)i * stride] : 0); }
    __device__ __forceinline__ int y(int i) const { return yb[(size_t)i * stride]; }
};

#endif  // DG_HOST_EMULATION

#ifdef DG_COOP
// LDS carve-up for a tree of n vertices with np coordinate entries: coordinates, k-d ordered ids, node results (heap order), triangles.
// The coordinates and the order come first: the on-GPU preparation (dg_prepare) produces the order in place and uses the region behind
// it - node results and triangles, not yet in use then - as its scratch.
struct DgLds {
    DG_LDS uint32_t *pxy;  // vertices: (x & 0xFFFF) | y << 16
    DG_LDS uint16_t *ord;
    DG_LDS uint32_t *res;  // [2 << depth]: farleft | farright << 16
    DG_LDS uint32_t *W;    // triangles, 3 words each
    __device__ __forceinline__ int vx(int i) const { return (int)(int16_t)(pxy[i] & 0xFFFFu); }
    __device__ __forceinline__ int vy(int i) const { return (int)pxy[i] >> 16; }
    __device__ __forceinline__ void set_vertex(int i, int x, int y) const { pxy[i] = ((uint32_t)x & 0xFFFFu) | ((uint32_t)y << 16); }
};
__host__ __device__ inline size_t dg_head_words(int n, int np) { return (size_t)np + ((size_t)n + 1) / 2; }  // coordinates, order in 32-bit words
__device__ __forceinline__ DgLds dg_carve(DG_LDS uint32_t *base, int n, int np) {
    const int depth = dg_depth(n);
    DgLds L;
    L.pxy = base;
    L.ord = (DG_LDS uint16_t *)(base + np);
    L.res = base + dg_head_words(n, np);
    L.W = L.res + (2 << depth);
    return L;
}
#endif  // DG_COOP

#ifndef DG_HOST_EMULATION

// Triangle list of a finished mesh, in pool order without the bounding triangles: corners org / dest / apex at orientation 0.
// vtx(t, k) reads corner k of slot t.  All DG_THREADS threads of the workgroup call it.
template <class F>
__device__ __forceinline__ void dg_emit(int nslots, F vtx, uint32_t ghost, int32_t *__restrict__ out, int32_t *__restrict__ count) {
    __shared__ int s_total;
    __shared__ int s_part[DG_THREADS];
    const int tid = threadIdx.x;
    const int per = (nslots + DG_THREADS - 1) / DG_THREADS;
    const int t0 = min(1 + tid * per, nslots), t1 = min(t0 + per, nslots);
    int mine = 0;
    for (int t = t0; t < t1; t++) mine += (vtx(t, 0) != ghost && vtx(t, 1) != ghost && vtx(t, 2) != ghost) ? 1 : 0;
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
        const uint32_t v0 = vtx(t, 0), v1 = vtx(t, 1), v2 = vtx(t, 2);
        if (v0 == ghost || v1 == ghost || v2 == ghost) continue;
        out[3 * pos] = (int32_t)v1;
        out[3 * pos + 1] = (int32_t)v2;
        out[3 * pos + 2] = (int32_t)v0;
        pos++;
    }
    if (tid == 0) *count = s_total;
}

// The tree of a set whose coordinates and k-d order are in LDS already (m <= DG_SUB_MAX): bottom-up, then the triangle list.
// clk (test hook, may be nullptr): clk[0] = start, clk[1 + d] = end of tree depth d in wall_clock64 ticks (100 MHz), for the first workgroup
template <bool NARROW>
__device__ __forceinline__ void dg_build_and_emit(const DgLds &L, int m, int32_t *__restrict__ out, int32_t *__restrict__ count, long long *clk = nullptr) {
    const int tid = threadIdx.x;
    if (tid == 0) L.W[0] = 0u, L.W[1] = GHOST << 16, L.W[2] = GHOST | (GHOST << 16);  // slot 0, "outer space"
    __syncthreads();
    if (clk && tid == 0 && blockIdx.x == 0) clk[0] = wall_clock64();
    const Mesh M{L.W, L.pxy};
    for (int d = dg_depth(m); d >= 0; d--) {
        for (int j = tid; j < (1 << d); j += DG_THREADS) d_process_node<NARROW>(M, L.res, L.ord, m, d, j);
        __syncthreads();
        if (clk && tid == 0 && blockIdx.x == 0) clk[1 + d] = wall_clock64();
    }
    DG_LDS const uint16_t *F = (DG_LDS const uint16_t *)L.W;
    dg_emit(2 * m - 1, [F](int t, int k) { return (uint32_t)F[6 * t + 3 + k]; }, GHOST, out, count);
}

// A whole set in LDS (m <= DG_SUB_MAX), k-d order from memory.
template <bool NARROW>
__device__ __forceinline__ void dg_triangulate(const DgSet &S, long long *clk = nullptr) {
    extern __shared__ uint32_t dg_lds[];
    const int tid = threadIdx.x, m = S.m, npts = S.npts;
    if (m < 3) {
        if (tid == 0) *S.count = 0;
        return;
    }
    const DgLds L = dg_carve((DG_LDS uint32_t *)dg_lds, m, npts);
    for (int i = tid; i < npts; i += DG_THREADS) L.set_vertex(i, S.x(i), S.y(i));
    for (int i = tid; i < m; i += DG_THREADS) L.ord[i] = (uint16_t)S.order[i];
    dg_build_and_emit<NARROW>(L, m, S.out, S.count, clk);
}

// Subtree j of the cut depth of a large set: triangulated in LDS with local vertex numbers (position in the k-d order), then
// written to the set's global mesh gT in the slot numbers and vertex ids of the whole set; gxy gets the vertices' coordinates,
// gres[2h], gres[2h + 1] the far-left / far-right handles of heap node h = 2^c + j.
template <bool NARROW>
__device__ __forceinline__ void dg_subtree(const DgSet &S, int sub_max, int j, GTri *__restrict__ gT, int32_t *__restrict__ gxy, uint32_t *__restrict__ gres) {
    extern __shared__ uint32_t dg_lds[];
    const int tid = threadIdx.x;
    const int c = dg_cut_depth(S.m, sub_max);
    if (c > DG_CUT_MAX || j >= (1 << c)) return;  // (deeper cuts than the node-result table holds: the launchers never hand such a set over)
    int lo, n, axis0;
    uint32_t slot0;
    if (!d_node(S.m, c, j, lo, n, slot0, axis0) || n < 2) return;  // (every node above the cut has more than sub_max >= 6 vertices)
    const DgLds L = dg_carve((DG_LDS uint32_t *)dg_lds, n, n);
    for (int i = tid; i < n; i += DG_THREADS) {
        const int id = S.order[lo + i];
        const int x = S.x(id), y = S.y(id);
        L.set_vertex(i, x, y);
        L.ord[i] = (uint16_t)i;
        gxy[2 * (size_t)id] = x;
        gxy[2 * (size_t)id + 1] = y;
    }
    if (tid == 0) L.W[0] = 0u, L.W[1] = GHOST << 16, L.W[2] = GHOST | (GHOST << 16);
    __syncthreads();
    const Mesh M{L.W, L.pxy};
    for (int d = dg_depth(n); d >= 0; d--) {
        for (int q = tid; q < (1 << d); q += DG_THREADS) d_process_node<NARROW>(M, L.res, L.ord, n, d, q, axis0);
        __syncthreads();
    }
    DG_LDS const uint16_t *F = (DG_LDS const uint16_t *)L.W;
    for (int t = 1 + tid; t < 2 * n - 1; t += DG_THREADS) {
        uint32_t *g = reinterpret_cast<uint32_t *>(gT + (slot0 + t - 1));  // plain stores: the merges run in the next kernel
#pragma unroll
        for (int k = 0; k < 3; k++) {
            g[k] = dg_global_handle(F[6 * t + k], slot0);
            const uint32_t v = F[6 * t + 3 + k];
            g[3 + k] = v == GHOST ? GHOST32 : (uint32_t)S.order[lo + v];
        }
    }
    if (tid == 0) {
        const uint32_t r = L.res[1];
        gres[2 * ((1 << c) + j)] = dg_global_handle(r & 0xFFFFu, slot0);
        gres[2 * ((1 << c) + j) + 1] = dg_global_handle(r >> 16, slot0);
        if (j == 0) {
            uint32_t *g = reinterpret_cast<uint32_t *>(gT);
            g[0] = g[1] = g[2] = 0;
            g[3] = g[4] = g[5] = GHOST32;
        }
    }
}

// The merges above the cut, depth by depth, one lane of the first wavefront per merge (a fence between depths: a lane sees what
// the lanes of the depth below wrote), then the triangle list by the whole workgroup.
template <bool NARROW>
__device__ __forceinline__ void dg_top(const DgSet &S, int sub_max, GTri *gT, const int32_t *gxy, uint32_t *gres) {
    const int tid = threadIdx.x, m = S.m;
    const int c = dg_cut_depth(m, sub_max);
    if (c > DG_CUT_MAX) {
        if (tid == 0) *S.count = 0;
        return;
    }
    const MeshG M{gT, gxy, 16u * (uint32_t)m + 4096u};
    if (tid < 64)
        for (int d = c - 1; d >= 0; d--) {
            if (tid < (1 << d)) dg_top_node<NARROW>(M, (DG_VOLATILE uint32_t *)gres, m, d, tid);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the lanes of this wavefront: what depth d wrote, depth d - 1 reads)
        }
    __syncthreads();
    dg_emit(2 * m - 1, [gT](int t, int k) { return gT[t].w[3 + k]; }, GHOST32, S.out, S.count);
}

#endif  // DG_HOST_EMULATION

#ifdef DG_COOP
// ---- preparation on the GPU: (x, y) order, duplicate scan, k-d order ----------------------------------------------------------
// What Delaunay::prepare (host_stage.cpp) does for the host triangulation - and what the reference does with a randomised quicksort,
// a duplicate scan and randomised quickselects (triangle.cpp:5183-5360, 5889-5903) - for a vertex set that lives on the support
// lattice of an image: every vertex is a cell (column x, lattice row of y) of a sparse 2-D grid, so
//   * the (x, y) order is the column-major enumeration of a bit map of occupied cells: a vertex's rank is the number of set bits in
//     front of its own (prefix sums over the bit map's words), the (y, x) order the same on the row-major bit map;
//   * two vertices in one cell are coincident points.  Which of them survives the reference's duplicate scan depends on the pivot
//     sequence of its quicksort (the library's LCG).  The case real images produce - 7 of the 21 kitti_mini frames - is the image
//     corner (W-1, 0) whose nearest support point has disparity 0: elas.cpp:258-259 then adds the right-image corner (W-1+0, 0, 0), the
//     SAME (u, v, d) triple as the left-image corner.  Coincident vertices with equal disparity are interchangeable - same coordinates
//     in both images, same plane equations; only the index that appears in the triangle lists differs, and no map depends on it - so
//     the lowest id is kept here.  Coincident vertices with DIFFERENT disparities (possible with lr_threshold > 2) are not: such a set
//     is reported and the host stage triangulates it the reference's way;
//   * the alternating cuts are index splits of one order and stable partitions of the other (see Delaunay::kd_order): all nodes of
//     one depth of the cut tree at once, one prefix sum over "goes left" flags per depth.
// Everything is in LDS, in the region the node results and the triangles use afterwards.
struct DgPrep {      // lattice geometry, from the image size (engine.cpp: delaunay_prep_dims)
    int xmin, cols;  // vertex columns: x - xmin in [0, cols)  (x = u or u - d, corner points included: -disp_max .. W - 1 + disp_max)
    int step, ylast; // row(y) = y / step; the image's last line ylast = H - 1 (corner points) gets a row of its own when it is no lattice row
    int rows;        // lattice rows + 1
    int W1, W2;      // 32-bit words per column of the column-major bit map / per row of the row-major one
    int bm_words;    // max(cols * W1, rows * W2)
};

constexpr int DG_DUP_MAX = 16;  // coincident vertices a set may have (beyond the first of each cell) before it is handed to the host

__host__ __device__ inline size_t dg_prep_scratch_words(const DgPrep &pp, int m) {  // bit map, word prefixes (u16), X, Y (u32), idx, xr (u16), chunk table, scan cells, duplicates, dropped flags
    return (size_t)pp.bm_words + ((size_t)pp.bm_words + 1) / 2 + 2 * (size_t)m + 2 * (((size_t)m + 1) / 2) + DG_THREADS + 16 + 2 * DG_DUP_MAX + ((size_t)m + 31) / 32;
}

constexpr int DG_PREP_CHUNK = 16;                          // consecutive positions a thread owns in the k-d passes
constexpr int DG_PREP_MAX = DG_THREADS * DG_PREP_CHUNK;    // 4096 vertices

__device__ __forceinline__ int dg_block_scan(int val, DG_LDS int *cells, int *total) {  // exclusive prefix sum over the workgroup's DG_THREADS values
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = val;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        incl += lane >= off ? o : 0;
    }
    if (lane == 63) cells[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < DG_THREADS / 64; w++) {
        const int c = cells[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

// Exclusive prefix sums of the set bits of the bit map's words: pm[w] = bits set in words [0, w).  Returns through *total the bit count.
__device__ __forceinline__ void dg_word_prefix(DG_LDS const uint32_t *bm, DG_LDS uint16_t *pm, int nwords, DG_LDS int *cells, int *total) {
    const int per = (nwords + DG_THREADS - 1) / DG_THREADS;
    const int w0 = min((int)threadIdx.x * per, nwords), w1 = min(w0 + per, nwords);
    int mine = 0;
    for (int w = w0; w < w1; w++) mine += __popc(bm[w]);
    int acc = dg_block_scan(mine, cells, total);
    for (int w = w0; w < w1; w++) {
        pm[w] = (uint16_t)acc;
        acc += __popc(bm[w]);
    }
    __syncthreads();
}

// pxy holds the npts vertices' coordinates, dsp[i * dstride] their disparities (nullptr: unknown).  Writes ord[0 .. m) = the ids of the
// m surviving vertices in k-d order and returns m, or returns -1 (uniformly) when the set has coincident points that are not
// interchangeable or does not fit the bit maps - the caller leaves such a set to the host.
__device__ __forceinline__ int dg_prepare(const DgLds &L, int npts, const DgPrep &pp, const int32_t *__restrict__ dsp, int dstride) {
    const int tid = threadIdx.x;
    int m = npts;
    DG_LDS uint32_t *bm = L.res;
    DG_LDS uint16_t *pm = (DG_LDS uint16_t *)(bm + pp.bm_words);
    DG_LDS uint32_t *X = bm + pp.bm_words + (pp.bm_words + 1) / 2, *Y = X + m;
    DG_LDS uint16_t *idx = (DG_LDS uint16_t *)(Y + m), *xr = idx + m + (m & 1);
    DG_LDS uint32_t *chunk = (DG_LDS uint32_t *)(xr + m + (m & 1));  // per thread: prefix of its chunk's flags | flag bits << 16
    DG_LDS int *cells = (DG_LDS int *)(chunk + DG_THREADS);           // [0..3] wavefront totals, [4] "unusable" flag, [5] coincident vertices seen, [6] vertices dropped,
                                                                      // [7] "coincident vertices with different disparities" (a word of its own: a flag that is tested after barrier N
                                                                      //     is never written between barrier N and barrier N + 1, so every test is uniform)
    DG_LDS int *dup_id = cells + 16, *dup_min = dup_id + DG_DUP_MAX;   // the vertices that found their cell taken; lowest id of each one's cell
    DG_LDS uint32_t *dropped = (DG_LDS uint32_t *)(dup_min + DG_DUP_MAX);  // bit i: vertex i is a coincident point that does not survive
    if (m > DG_PREP_MAX || m > 0xFFFF) return -1;  // (uniform: an argument)
    if (tid == 0) cells[4] = 0, cells[5] = 0, cells[6] = 0, cells[7] = 0;
    for (int w = tid; w < (m + 31) / 32; w += DG_THREADS) dropped[w] = 0u;
    // ---- (x, y) ranks: column-major bit map
    for (int w = tid; w < pp.bm_words; w += DG_THREADS) bm[w] = 0u;
    __syncthreads();
    for (int i = tid; i < m; i += DG_THREADS) {
        const int x = L.vx(i) - pp.xmin, y = L.vy(i);
        const int row = (y == pp.ylast && pp.ylast % pp.step != 0) ? pp.rows - 1 : y / pp.step;
        if (x < 0 || x >= pp.cols || y < 0 || row >= pp.rows) {
            cells[4] = 1;
            continue;
        }
        const uint32_t bit = 1u << (row & 31);
        if (__hip_atomic_fetch_or(&bm[x * pp.W1 + (row >> 5)], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) {  // two vertices in one cell
            const int q = __hip_atomic_fetch_add(&cells[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (q < DG_DUP_MAX && dsp)
                dup_id[q] = i, dup_min[q] = i;
            else
                cells[4] = 1;
        }
    }
    __syncthreads();
    if (cells[4]) return -1;  // (uniform: read after the barrier, not written again)
    const int ndup = cells[5];
    if (ndup > 0) {
        // every vertex that shares a cell with one of the listed vertices is a member of that coincident group: the group's lowest id
        // survives, provided all members carry the same disparity (then they are the same support point twice)
        for (int i = tid; i < m; i += DG_THREADS)
            for (int q = 0; q < ndup; q++) {
                const int c = dup_id[q];
                if (L.pxy[i] == L.pxy[c]) {
                    if (dsp[(size_t)i * dstride] != dsp[(size_t)c * dstride]) cells[7] = 1;
                    __hip_atomic_fetch_min(&dup_min[q], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        __syncthreads();
        if (cells[7]) return -1;
        for (int i = tid; i < m; i += DG_THREADS)
            for (int q = 0; q < ndup; q++) {
                const int c = dup_id[q];
                if (L.pxy[i] == L.pxy[c] && i != dup_min[q]) {
                    if (!(__hip_atomic_fetch_or(&dropped[i >> 5], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & (1u << (i & 31))))
                        __hip_atomic_fetch_add(&cells[6], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
            }
        __syncthreads();
    }
    const int nall = m;  // vertices with coordinates; m: the survivors
    m = nall - cells[6];
    int total;
    dg_word_prefix(bm, pm, pp.cols * pp.W1, cells, &total);
    for (int i = tid; i < nall; i += DG_THREADS) {
        if ((dropped[i >> 5] >> (i & 31)) & 1u) continue;
        const int x = L.vx(i) - pp.xmin, y = L.vy(i);
        const int row = (y == pp.ylast && pp.ylast % pp.step != 0) ? pp.rows - 1 : y / pp.step;
        const int w = x * pp.W1 + (row >> 5);
        const int r = (int)pm[w] + __popc(bm[w] & ((1u << (row & 31)) - 1u));
        xr[i] = (uint16_t)r;
        idx[r] = (uint16_t)i;
    }
    __syncthreads();
    // ---- (y, x) ranks: row-major bit map in the same memory
    for (int w = tid; w < pp.bm_words; w += DG_THREADS) bm[w] = 0u;
    __syncthreads();
    for (int i = tid; i < nall; i += DG_THREADS) {
        const int x = L.vx(i) - pp.xmin, y = L.vy(i);
        const int row = (y == pp.ylast && pp.ylast % pp.step != 0) ? pp.rows - 1 : y / pp.step;
        __hip_atomic_fetch_or(&bm[row * pp.W2 + (x >> 5)], 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    dg_word_prefix(bm, pm, pp.rows * pp.W2, cells, &total);
    for (int i = tid; i < nall; i += DG_THREADS) {
        if ((dropped[i >> 5] >> (i & 31)) & 1u) continue;
        const int x = L.vx(i) - pp.xmin, y = L.vy(i);
        const int row = (y == pp.ylast && pp.ylast % pp.step != 0) ? pp.rows - 1 : y / pp.step;
        const int w = row * pp.W2 + (x >> 5);
        const uint32_t yrank = (uint32_t)pm[w] + (uint32_t)__popc(bm[w] & ((1u << (x & 31)) - 1u));
        const uint32_t e = (yrank << 16) | (uint32_t)xr[i];  // a vertex as (rank by (y, x)) << 16 | rank by (x, y)
        X[xr[i]] = e;
        Y[yrank] = e;
    }
    __syncthreads();
    // ---- alternating cuts, one depth of the tree per pass.  A thread owns DG_PREP_CHUNK consecutive POSITIONS; the node a position
    // belongs to is a range [lo, lo + n) that halves from pass to pass (n >> 1 to the left, the rest to the right, like the recursion).
    int lo[DG_PREP_CHUNK], nn[DG_PREP_CHUNK];
#pragma unroll
    for (int k = 0; k < DG_PREP_CHUNK; k++) lo[k] = 0, nn[k] = m;
    const int p0 = tid * DG_PREP_CHUNK;
    const int depth = dg_depth(m);
    for (int dd = 0; dd < depth; dd++) {
        const bool by_x = (dd & 1) == 0;                      // cut by x: X splits by index, Y is partitioned (and the other way round)
        DG_LDS uint32_t *src = by_x ? Y : X;
        DG_LDS const uint32_t *ref = by_x ? X : Y;
        uint32_t e[DG_PREP_CHUNK];
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < DG_PREP_CHUNK; k++) {
            const int i = p0 + k;
            e[k] = 0;
            if (i < m) {
                e[k] = src[i];
                if (nn[k] > 3) {
                    const uint32_t pe = ref[lo[k] + (nn[k] >> 1)];
                    const uint32_t pivot = by_x ? (pe & 0xFFFFu) : (pe >> 16), r = by_x ? (e[k] & 0xFFFFu) : (e[k] >> 16);
                    bits |= (r < pivot ? 1u : 0u) << k;
                }
            }
        }
        const int base = dg_block_scan(__popc(bits), cells, &total);
        chunk[tid] = (uint32_t)base | (bits << 16);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DG_PREP_CHUNK; k++) {
            const int i = p0 + k;
            if (i < m && nn[k] > 3) {
                const uint32_t c0 = chunk[lo[k] / DG_PREP_CHUNK];
                const int at_lo = (int)(c0 & 0xFFFFu) + __popc((c0 >> 16) & ((1u << (lo[k] % DG_PREP_CHUNK)) - 1u));
                const int before = base + __popc(bits & ((1u << k) - 1u)) - at_lo;  // "goes left" flags of this node in front of position i
                const int nl = nn[k] >> 1;
                const int dest = ((bits >> k) & 1u) ? lo[k] + before : lo[k] + nl + (i - lo[k] - before);
                src[dest] = e[k];  // in place: every element of the pass has been read (barrier above)
                if (i >= lo[k] + nl) {
                    lo[k] += nl;
                    nn[k] -= nl;
                } else {
                    nn[k] = nl;
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < m; i += DG_THREADS) L.ord[i] = idx[X[i] & 0xFFFFu];  // leaves are in (x, y) order: the X arrangement is the result
    __syncthreads();
    return m;
}

// ---- the same preparation for sets beyond LDS (4K lattices: 21 000 - 30 000 vertices): one workgroup of 1 024 threads per set, bit
// maps, rank prefixes and the two vertex orders in the slot's global-memory scratch (the workgroup runs on one CU and is the only one that
// touches its set's scratch: workgroup-scope accesses, the phases ordered by __syncthreads()).  Same steps as dg_prepare; a position's node
// is found by walking the halving tree from the root each pass (no per-position state in registers).
constexpr int DGP_THREADS = 1024;

struct DgPrepScratch {  // per set: bit map, word prefixes, three vertex-order arrays (X, Y, spare), idx, xr, pref - all 32-bit words
    uint32_t *base;
    int cap;       // vertices per set the arrays hold
    int bm_words;  // words of the larger bit map
    __host__ __device__ size_t words_per_set() const { return 2 * (size_t)bm_words + 6 * (size_t)cap; }
    __device__ __forceinline__ uint32_t *set(int s) const { return base + (size_t)s * words_per_set(); }
};

// (workgroup scope: the set's one workgroup runs on one CU, whose vector cache all its wavefronts share; __syncthreads() orders the phases)
__device__ __forceinline__ uint32_t gld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void gst(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ int dgp_block_scan(int val, int *cells, int *total) {  // exclusive prefix sum over DGP_THREADS values (cells: LDS, 16 ints)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = val;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        incl += lane >= off ? o : 0;
    }
    if (lane == 63) cells[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < DGP_THREADS / 64; w++) {
        const int c = cells[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

__device__ __forceinline__ void dgp_word_prefix(const uint32_t *bm, uint32_t *pm, int nwords, int *cells) {
    const int per = (nwords + DGP_THREADS - 1) / DGP_THREADS;
    const int w0 = min((int)threadIdx.x * per, nwords), w1 = min(w0 + per, nwords);
    int mine = 0, total;
    for (int w = w0; w < w1; w++) mine += __popc(gld(bm + w));
    int acc = dgp_block_scan(mine, cells, &total);
    for (int w = w0; w < w1; w++) {
        gst(pm + w, (uint32_t)acc);
        acc += __popc(gld(bm + w));
    }
    __syncthreads();
}

// sup: the set's (u, v, d) triples; side: 0 left image (x = u), 1 right image (x = u - d).  Writes ord_out[0] = m and ord_out[1 .. m] = the
// ids in k-d order, or ord_out[0] = -1 (coincident points that are not interchangeable, vertices outside the bit maps).  All
// DGP_THREADS threads call it; returns m or -1 uniformly.
__device__ __forceinline__ int dg_prepare_global(const int32_t *__restrict__ sup, int npts, int side, const DgPrep &pp, uint32_t *scr, int cap, int bm_words, int32_t *ord_out) {
    __shared__ int cells[24];            // [0..15] wavefront totals, [16] unusable, [17] coincident vertices seen, [18] dropped, [19] coincident vertices with different
                                         // disparities (its own word: no flag is written between the barrier it is tested after and the next one)
    __shared__ int dup_id[DG_DUP_MAX], dup_min[DG_DUP_MAX];
    const int tid = threadIdx.x;
    uint32_t *bm = scr, *pm = bm + bm_words, *X = pm + bm_words, *Y = X + cap, *Z = Y + cap, *idx = Z + cap, *xr = idx + cap, *pref = xr + cap;
    if (npts > cap || npts > 0xFFFF) {  // (uniform: arguments)
        if (tid == 0) ord_out[0] = -1;
        return -1;
    }
    if (tid == 0) cells[16] = 0, cells[17] = 0, cells[18] = 0, cells[19] = 0;
    auto vx = [&](int i) { return sup[3 * i] - (side ? sup[3 * i + 2] : 0) - pp.xmin; };
    auto vrow = [&](int i) {
        const int y = sup[3 * i + 1];
        return y < 0 ? pp.rows : ((y == pp.ylast && pp.ylast % pp.step != 0) ? pp.rows - 1 : y / pp.step);
    };
    for (int w = tid; w < bm_words; w += DGP_THREADS) gst(bm + w, 0u);
    for (int i = tid; i < npts; i += DGP_THREADS) gst(pref + i, 0u);  // doubles as the "dropped" flags until the first pass
    __syncthreads();
    // ---- (x, y) ranks: column-major bit map; coincident vertices
    for (int i = tid; i < npts; i += DGP_THREADS) {
        const int x = vx(i), row = vrow(i);
        if (x < 0 || x >= pp.cols || row >= pp.rows) {
            cells[16] = 1;
            continue;
        }
        const uint32_t bit = 1u << (row & 31);
        if (__hip_atomic_fetch_or(bm + x * pp.W1 + (row >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) {
            const int q = atomicAdd(&cells[17], 1);
            if (q < DG_DUP_MAX)
                dup_id[q] = i, dup_min[q] = i;
            else
                cells[16] = 1;
        }
    }
    __syncthreads();
    const int ndup = cells[17];
    if (!cells[16] && ndup > 0) {  // a coincident group survives as its lowest id when all its members carry the same disparity
        for (int i = tid; i < npts; i += DGP_THREADS)
            for (int q = 0; q < ndup; q++) {
                const int c = dup_id[q];
                if (vx(i) == vx(c) && sup[3 * i + 1] == sup[3 * c + 1]) {
                    if (sup[3 * i + 2] != sup[3 * c + 2]) cells[19] = 1;
                    atomicMin(&dup_min[q], i);
                }
            }
        __syncthreads();
        for (int i = tid; i < npts; i += DGP_THREADS)
            for (int q = 0; q < ndup; q++) {
                const int c = dup_id[q];
                if (vx(i) == vx(c) && sup[3 * i + 1] == sup[3 * c + 1] && i != dup_min[q]) {
                    if (__hip_atomic_exchange(pref + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) atomicAdd(&cells[18], 1);
                    break;
                }
            }
    }
    __syncthreads();
    if (cells[16] || cells[19]) {
        if (tid == 0) ord_out[0] = -1;
        return -1;
    }
    const int m = npts - cells[18];
    dgp_word_prefix(bm, pm, pp.cols * pp.W1, cells);
    for (int i = tid; i < npts; i += DGP_THREADS) {
        if (gld(pref + i)) continue;  // dropped
        const int x = vx(i), row = vrow(i), w = x * pp.W1 + (row >> 5);
        const uint32_t r = gld(pm + w) + (uint32_t)__popc(gld(bm + w) & ((1u << (row & 31)) - 1u));
        gst(xr + i, r);
        gst(idx + r, (uint32_t)i);
    }
    __syncthreads();
    // ---- (y, x) ranks: row-major bit map in the same memory
    for (int w = tid; w < bm_words; w += DGP_THREADS) gst(bm + w, 0u);
    __syncthreads();
    for (int i = tid; i < npts; i += DGP_THREADS) {
        const int x = vx(i), row = vrow(i);
        __hip_atomic_fetch_or(bm + row * pp.W2 + (x >> 5), 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    dgp_word_prefix(bm, pm, pp.rows * pp.W2, cells);
    for (int i = tid; i < npts; i += DGP_THREADS) {
        if (gld(pref + i)) continue;
        const int x = vx(i), row = vrow(i), w = row * pp.W2 + (x >> 5);
        const uint32_t yrank = gld(pm + w) + (uint32_t)__popc(gld(bm + w) & ((1u << (x & 31)) - 1u)), xrank = gld(xr + i);
        const uint32_t e = (yrank << 16) | xrank;
        gst(X + xrank, e);
        gst(Y + yrank, e);
    }
    __syncthreads();
    // ---- alternating cuts, one depth per pass; the partitioned order goes to the spare array, which then takes its place.  Inside the
    // passes the arrays are read with plain loads (the compiler keeps many in flight); what other threads of this workgroup wrote in the
    // pass before is visible after the barrier (same CU, same vector cache).  An agent-scope fence would write back the XCD's L2 - full
    // of the other streams' maps - four times per pass.
    __syncthreads();
    const int depth = dg_depth(m);
    const int per = (m + DGP_THREADS - 1) / DGP_THREADS, c0 = min(tid * per, m), c1 = min(c0 + per, m);
    uint32_t *ax = X, *ay = Y, *spare = Z;
    for (int dd = 0; dd < depth; dd++) {
        const bool by_x = (dd & 1) == 0;
        const uint32_t *src = by_x ? ay : ax, *ref = by_x ? ax : ay;
        int mine = 0;
        for (int i = c0; i < c1; i++) {
            int lo = 0, n = m;
            for (int b = 0; b < dd && n > 3; b++) {  // the node of depth dd this position belongs to (or the leaf above it)
                const int nl = n >> 1;
                if (i >= lo + nl)
                    lo += nl, n -= nl;
                else
                    n = nl;
            }
            uint32_t flag = 0;
            if (n > 3) {  // (a leaf reached before depth dd keeps n <= 3: not split)
                const uint32_t pe = ref[lo + (n >> 1)], e = src[i];
                const uint32_t pivot = by_x ? (pe & 0xFFFFu) : (pe >> 16), r = by_x ? (e & 0xFFFFu) : (e >> 16);
                flag = r < pivot ? 1u : 0u;
            }
            pref[i] = flag << 31;  // (own positions only until the scan below)
            mine += (int)flag;
        }
        int total;
        int acc = dgp_block_scan(mine, cells, &total);
        for (int i = c0; i < c1; i++) {
            const uint32_t f = pref[i] >> 31;
            pref[i] = (f << 31) | (uint32_t)acc;
            acc += (int)f;
        }
        __syncthreads();
        for (int i = c0; i < c1; i++) {
            int lo = 0, n = m;
            for (int b = 0; b < dd && n > 3; b++) {
                const int nl = n >> 1;
                if (i >= lo + nl)
                    lo += nl, n -= nl;
                else
                    n = nl;
            }
            const uint32_t e = src[i];
            int dest = i;
            if (n > 3) {
                const uint32_t pi = pref[i], pl = pref[lo];
                const int before = (int)(pi & 0x7FFFFFFFu) - (int)(pl & 0x7FFFFFFFu), nl = n >> 1;
                dest = (pi >> 31) ? lo + before : lo + nl + (i - lo - before);
            }
            spare[dest] = e;
        }
        __syncthreads();
        if (by_x) {
            uint32_t *t = ay;
            ay = spare, spare = t;
        } else {
            uint32_t *t = ax;
            ax = spare, spare = t;
        }
    }
    for (int i = tid; i < m; i += DGP_THREADS) ord_out[1 + i] = (int32_t)idx[ax[i] & 0xFFFFu];
    if (tid == 0) ord_out[0] = m;
    return m;
}
#endif  // DG_COOP

#ifndef DG_HOST_EMULATION

// Pipeline form for resident chunks: the sets of more than sub_max vertices (up to large_cap) of a chunk whose blob k_delaunay_resident
// has laid out; leaves [m, ids ...] where the host stage would have left them, for k_dgl_subtrees_blob / k_dgl_top_blob.
__global__ __launch_bounds__(DGP_THREADS) void k_dg_prepare_large_blob(int32_t *__restrict__ blob, int sub_max, int large_cap, DgPrep pp, DgPrepScratch sc) {
    const int set = blockIdx.x, pair = set >> 1, side = set & 1;
    int32_t *meta = blob + (size_t)pair * META_WORDS;
    const int ns = meta[0];
    if (ns <= sub_max || ns > large_cap || meta[7] == 0) return;
    int32_t *ord = blob + meta[5] + (size_t)2 * ns * 3 + (size_t)side * (ns + 1);
    const int m = dg_prepare_global(blob + meta[1], ns, side, pp, sc.set(set), sc.cap, sc.bm_words, ord);
    if (m < 0 && threadIdx.x == 0) meta[2 + 2 * side] = -1;  // the host stage builds this side
}


// Test-hook form: sets[s] = {offset of the set's first entry in `order` / `xy`, m (vertices after the duplicate scan), n_points
// (entries of xy), offset of its triangle list in tri_out}.  order: vertex ids in k-d order; xy: (x, y) per id.
__device__ __forceinline__ DgSet dg_set_from_list(const int4 st, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count) {
    DgSet S;
    S.m = st.y, S.npts = st.z;
    S.order = order + st.x;
    S.xb = xy + 2 * (size_t)st.x, S.yb = S.xb + 1, S.db = nullptr, S.stride = 2;
    S.out = tri_out + st.w, S.count = tri_count;
    return S;
}

// Pipeline form: set = pair * 2 + side.  The chunk's blob (engine.cpp) holds per pair the meta words, the support points
// (u, v, d) and room for the two triangle lists; behind the second list the host stage has left, per side, [m, id_0 .. id_{m-1}]:
// the k-d ordered vertex ids (m = -1: the host triangulated this side itself, nothing to do).  Left image: (u, v); right: (u - d, v).
// False: nothing to do for this set here (sets of more than sub_max points belong to the k_dgl_* kernels: `large`).
__device__ __forceinline__ bool dg_set_from_blob(int32_t *blob, int set, int sub_max, bool large, DgSet &S) {
    const int pair = set >> 1, side = set & 1;
    int32_t *meta = blob + (size_t)pair * META_WORDS;
    const int ns = meta[0];
    if (ns < 3 || meta[7] == 0 || (ns > sub_max) != large) return false;  // meta[7]: this pair is triangulated here (the host pool did the others)
    const int32_t *sup = blob + meta[1];
    const int32_t *ord = blob + meta[5] + (size_t)2 * ns * 3 + (size_t)side * (ns + 1);
    if (ord[0] < 0) return false;
    S.m = ord[0], S.npts = ns;
    S.order = ord + 1;
    S.xb = sup, S.yb = sup + 1, S.db = side ? sup + 2 : nullptr, S.stride = 3;
    S.out = blob + meta[3 + 2 * side], S.count = meta + 2 + 2 * side;
    return true;
}

// narrow: every coordinate difference of every set is below 2^14 in magnitude (lv_incirc)
__global__ __launch_bounds__(DG_THREADS) void k_delaunay(const int4 *__restrict__ sets, const int32_t *__restrict__ order, const int32_t *__restrict__ xy,
                                                        int32_t *__restrict__ tri_out, int32_t *__restrict__ tri_count, int narrow, long long *clk) {
    const DgSet S = dg_set_from_list(sets[blockIdx.x], order, xy, tri_out, tri_count + blockIdx.x);
    if (narrow)
        dg_triangulate<true>(S, clk);
    else
        dg_triangulate<false>(S, clk);
}

__global__ __launch_bounds__(DG_THREADS) void k_delaunay_blob(int32_t *__restrict__ blob, int sub_max) {
    DgSet S;
    if (dg_set_from_blob(blob, blockIdx.x, sub_max, false, S)) dg_triangulate<true>(S);  // (the engine's sets are narrow)
}

// The resident form: the chunk's support-point lists never leave the device.  set = pair * 2 + side; `fsup` / `fnsup` are the lattice
// filter's output ([cap][max_pts][3] triples (u, v, d), [cap] counts).  The kernel lays the pair's part of the blob out itself - at a
// fixed place per pair, with the offsets the host stage would have chosen inside it - copies the support points there, prepares the
// vertex set (dg_prepare) and triangulates it.  Meta words (engine.cpp): [0] points, [1] their offset, [2]/[4] triangles left / right,
// [3]/[5] their offsets, [7] = 1.  A side this kernel cannot do - coincident points, or more than sub_max vertices (what the launch has
// LDS for) - gets its triangle count set to -1: the host stage builds it and phase 2 uploads the list.
__global__ __launch_bounds__(DG_THREADS) void k_delaunay_resident(const int32_t *__restrict__ fsup, const int32_t *__restrict__ fnsup, int32_t *__restrict__ blob, int cap, int max_pts,
                                                                 int pair_words, int sub_max, int large_min, int large_cap, DgPrep pp) {
    extern __shared__ uint32_t dg_lds[];
    const int tid = threadIdx.x, pair = blockIdx.x >> 1, side = blockIdx.x & 1;
    int32_t *meta = blob + (size_t)pair * META_WORDS;
    int ns = fnsup[pair];
    if (ns < 0 || ns > max_pts) ns = 0;  // (never: the filter counts what it wrote)
    const int off_sup = cap * META_WORDS + pair * pair_words, off_t1 = off_sup + 3 * ns, off_t2 = off_t1 + 6 * ns;
    if (side == 0 && tid == 0) {
        meta[0] = ns;
        meta[1] = off_sup;
        meta[3] = off_t1;
        meta[5] = off_t2;
        meta[6] = 0;
        meta[7] = 1;
    }
    if (ns < 3) {
        if (tid == 0) meta[2 + 2 * side] = 0;
        return;
    }
    const int32_t *sup = fsup + (size_t)pair * max_pts * 3;
    if (side == 0)
        for (int i = tid; i < 3 * ns; i += DG_THREADS) blob[off_sup + i] = sup[i];
    if (ns > sub_max) {
        // more vertices than this launch has LDS for: a set of the cut path's size (large_min < ns <= large_cap) is prepared and
        // triangulated by the kernels behind this one (k_dg_prepare_large_blob, k_dgl_*_blob), anything else goes back to the host stage
        if (tid == 0) meta[2 + 2 * side] = (ns > large_min && ns <= large_cap) ? 0 : -1;
        return;
    }
    const DgLds L = dg_carve((DG_LDS uint32_t *)dg_lds, ns, ns);
    for (int i = tid; i < ns; i += DG_THREADS) {
        const int u = sup[3 * i], v = sup[3 * i + 1], dsp = sup[3 * i + 2];
        L.set_vertex(i, side ? u - dsp : u, v);  // elas.cpp:449-461: left image (u, v), right image (u - d, v)
    }
    __syncthreads();
    const int m = dg_prepare(L, ns, pp, sup + 2, 3);
    if (m < 0) {
        if (tid == 0) meta[2 + 2 * side] = -1;
        return;
    }
    dg_build_and_emit<true>(L, m, blob + (side ? off_t2 : off_t1), meta + 2 + 2 * side);  // (narrow: columns within [-disp_max, W + disp_max), W <= 8192)
}

// Test hook: only the preparation of one vertex set given as (x, y) pairs; ord_out[0] = m or -1, then the ids in k-d order.
__global__ __launch_bounds__(DG_THREADS) void k_dg_prepare_test(const int32_t *__restrict__ xy, const int32_t *__restrict__ dsp, int n, int32_t *__restrict__ ord_out, DgPrep pp) {
    extern __shared__ uint32_t dg_lds[];
    const int tid = threadIdx.x;
    const DgLds L = dg_carve((DG_LDS uint32_t *)dg_lds, n, n);
    for (int i = tid; i < n; i += DG_THREADS) L.set_vertex(i, xy[2 * i], xy[2 * i + 1]);
    __syncthreads();
    const int m = dg_prepare(L, n, pp, dsp, 1);
    if (tid == 0) ord_out[0] = m;
    if (m > 0)
        for (int i = tid; i < m; i += DG_THREADS) ord_out[1 + i] = L.ord[i];
}

// Large sets: scratch per set = GTri[2 * cap], (x, y)[cap], node results [4 << DG_CUT_MAX] (cap: the largest set the scratch holds)
struct DgScratch {
    GTri *tri;
    int32_t *xy;
    uint32_t *res;
    int cap;
    __device__ __forceinline__ GTri *T(int set) const { return tri + (size_t)set * 2 * cap; }
    __device__ __forceinline__ int32_t *XY(int set) const { return xy + (size_t)set * 2 * cap; }
    __device__ __forceinline__ uint32_t *R(int set) const { return res + (size_t)set * (4 << DG_CUT_MAX); }
};

__global__ __launch_bounds__(DG_THREADS) void k_dgl_subtrees(const int4 *__restrict__ sets, const int32_t *__restrict__ order, const int32_t *__restrict__ xy,
                                                            int32_t *__restrict__ tri_out, int32_t *__restrict__ tri_count, int sub_max, DgScratch sc, int narrow) {
    const int set = blockIdx.y;
    const DgSet S = dg_set_from_list(sets[set], order, xy, tri_out, tri_count + set);
    if (narrow)
        dg_subtree<true>(S, sub_max, blockIdx.x, sc.T(set), sc.XY(set), sc.R(set));
    else
        dg_subtree<false>(S, sub_max, blockIdx.x, sc.T(set), sc.XY(set), sc.R(set));
}

__global__ __launch_bounds__(DG_THREADS) void k_dgl_top(const int4 *__restrict__ sets, const int32_t *__restrict__ order, const int32_t *__restrict__ xy,
                                                       int32_t *__restrict__ tri_out, int32_t *__restrict__ tri_count, int sub_max, DgScratch sc, int narrow) {
    const int set = blockIdx.x;
    const DgSet S = dg_set_from_list(sets[set], order, xy, tri_out, tri_count + set);
    if (narrow)
        dg_top<true>(S, sub_max, sc.T(set), sc.XY(set), sc.R(set));
    else
        dg_top<false>(S, sub_max, sc.T(set), sc.XY(set), sc.R(set));
}

__global__ __launch_bounds__(DG_THREADS) void k_dgl_subtrees_blob(int32_t *__restrict__ blob, int sub_max, DgScratch sc) {
    const int set = blockIdx.y;
    DgSet S;
    if (dg_set_from_blob(blob, set, sub_max, true, S) && S.npts <= sc.cap) dg_subtree<true>(S, sub_max, blockIdx.x, sc.T(set), sc.XY(set), sc.R(set));
}

__global__ __launch_bounds__(DG_THREADS) void k_dgl_top_blob(int32_t *__restrict__ blob, int sub_max, DgScratch sc) {
    const int set = blockIdx.x;
    DgSet S;
    if (dg_set_from_blob(blob, set, sub_max, true, S) && S.npts <= sc.cap) dg_top<true>(S, sub_max, sc.T(set), sc.XY(set), sc.R(set));
}

#endif  // DG_HOST_EMULATION

}  // namespace dg

using namespace dg;

#ifndef DG_HOST_EMULATION
size_t delaunay_gpu_lds_bytes(int m, int npts) {
    const size_t nslots = 2 * (size_t)m - 1;
    return sizeof(uint32_t) * (dg_head_words(m, npts) + (2 << dg_depth(m))) + sizeof(DTri) * nslots + 16;
}

#endif  // DG_HOST_EMULATION

#ifdef DG_COOP
// Lattice geometry of the on-GPU preparation for images of W x H with lattice step `step` and disparities up to disp_max.
static DgPrep dg_prep_dims(int W, int H, int step, int disp_max) {
    DgPrep pp;
    pp.xmin = -disp_max;
    pp.cols = W + 2 * disp_max + 1;
    pp.step = step;
    pp.ylast = H - 1;
    pp.rows = (H + step - 1) / step + 1;
    pp.W1 = (pp.rows + 31) / 32;
    pp.W2 = (pp.cols + 31) / 32;
    pp.bm_words = pp.cols * pp.W1 > pp.rows * pp.W2 ? pp.cols * pp.W1 : pp.rows * pp.W2;
    return pp;
}

// LDS of the resident kernel for sets of up to m vertices: coordinates and order, then the larger of the preparation's scratch and
// the triangulation's node results + triangles
size_t delaunay_resident_lds_bytes(int W, int H, int step, int disp_max, int m) {
    const DgPrep pp = dg_prep_dims(W, H, step, disp_max);
    const size_t nslots = 2 * (size_t)m - 1;
    const size_t tri = sizeof(uint32_t) * (2 << dg_depth(m)) + sizeof(DTri) * nslots;
    const size_t prep = sizeof(uint32_t) * dg_prep_scratch_words(pp, m);
    return sizeof(uint32_t) * dg_head_words(m, m) + (tri > prep ? tri : prep) + 16;
}
int delaunay_prep_max_points() { return DG_PREP_MAX; }
#endif  // DG_COOP

#ifndef DG_HOST_EMULATION

// Both triangulations of every pair of a chunk from the lattice filter's lists on the device (k_delaunay_resident).  ns_max: an upper
// bound of the chunk's support counts that take the LDS path (sizes the LDS request).
void launch_delaunay_resident(const int32_t *fsup, const int32_t *fnsup, int32_t *blob, int cap, int max_pts, int pair_words, int n_pairs, int ns_max, int sub_max, int W, int H, int step,
                              int disp_max, hipStream_t st, int large_min, int large_cap) {
    // (ns_max < 3: no set of this launch takes the LDS path - wide images whose bit maps do not fit LDS - the kernel only lays the blob out)
    const size_t lds = ns_max < 3 ? 1024 : delaunay_resident_lds_bytes(W, H, step, disp_max, ns_max);
    static std::atomic<size_t> granted[64];
    ensure_dynamic_lds(k_delaunay_resident, lds, granted, "delaunay_gpu (resident)");
    SV_LAUNCH(K_DELAUNAY, k_delaunay_resident, dim3(2 * n_pairs), dim3(DG_THREADS), lds, st, fsup, fnsup, blob, cap, max_pts, pair_words, sub_max, large_min, large_cap,
              dg_prep_dims(W, H, step, disp_max));
}

// Scratch of the large-set preparation: bytes per vertex set for sets of up to `cap` vertices on the lattice of a W x H image
size_t delaunay_prep_large_bytes(int W, int H, int step, int disp_max, int cap) {
    const DgPrep pp = dg_prep_dims(W, H, step, disp_max);
    return sizeof(uint32_t) * (2 * (size_t)pp.bm_words + 6 * (size_t)cap);
}

// A set of more than sub_max << DG_CUT_MAX vertices needs more subtrees than a set's node-result table holds (round 4: a 23 000-vertex
// 4K list with sub_max = 350 overran it).  sv_create keeps the handle's limit below that and the kernels leave deeper sets alone;
// a caller that asks for more gets an error, not a silently smaller launch.
static void dg_check_cut(int ns_max, int sub_max, const DelaunayScratch &scratch) {
    if (sub_max < 6 || sub_max > DG_SUB_MAX || dg_cut_depth(ns_max, sub_max) > DG_CUT_MAX)
        throw std::invalid_argument("delaunay_gpu: sets of " + std::to_string(ns_max) + " vertices cannot be cut into " + std::to_string(1 << DG_CUT_MAX) + " subtrees of " +
                                    std::to_string(sub_max));
    if (ns_max > scratch.cap) throw std::invalid_argument("delaunay_gpu: the scratch holds sets of " + std::to_string(scratch.cap) + " vertices, asked for " + std::to_string(ns_max));
}

// The sets of a resident chunk that are beyond LDS (sub_max < vertices <= large_cap): preparation, then the cut path's two kernels.
// prep_scratch: 2 * n_pairs sets of delaunay_prep_large_bytes(.., scratch.cap).
void launch_delaunay_resident_large(int32_t *blob, int n_pairs, int sub_max, int large_cap, int W, int H, int step, int disp_max, void *prep_scratch, const DelaunayScratch &scratch,
                                    hipStream_t st) {
    dg_check_cut(large_cap < scratch.cap ? large_cap : scratch.cap, sub_max, scratch);
    const DgPrep pp = dg_prep_dims(W, H, step, disp_max);
    const DgPrepScratch ps{static_cast<uint32_t *>(prep_scratch), scratch.cap, pp.bm_words};
    SV_LAUNCH(K_DELAUNAY, k_dg_prepare_large_blob, dim3(2 * n_pairs), dim3(DGP_THREADS), 0, st, blob, sub_max, large_cap < scratch.cap ? large_cap : scratch.cap, pp, ps);
    launch_delaunay_blob_large(blob, n_pairs, large_cap < scratch.cap ? large_cap : scratch.cap, sub_max, scratch, st);
}

// Test hook: the preparation alone for one set of (x, y) pairs on the lattice of a W x H image; ord_out[0] = m or -1, then m ids
int launch_delaunay_prepare_test(const int32_t *d_xy, const int32_t *d_dsp, int n, int32_t *d_ord_out, int W, int H, int step, int disp_max, hipStream_t st) {
    const size_t lds = delaunay_resident_lds_bytes(W, H, step, disp_max, n < 3 ? 3 : n);
    if (lds > 160 * 1024 || n > DG_PREP_MAX) return -1;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_dg_prepare_test), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_dg_prepare_test, dim3(1), dim3(DG_THREADS), lds, st, d_xy, d_dsp, n, d_ord_out, dg_prep_dims(W, H, step, disp_max));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int delaunay_gpu_max_points() { return DG_SUB_MAX; }
int delaunay_gpu_cut_max() { return DG_CUT_MAX; }
int delaunay_gpu_large_max_points() { return DG_SUB_MAX << DG_CUT_MAX; }

size_t delaunay_scratch_bytes(int cap, int nsets, size_t *tri_bytes, size_t *xy_bytes, size_t *res_bytes) {
    const size_t t = sizeof(GTri) * 2 * (size_t)cap * nsets, x = sizeof(int32_t) * 2 * (size_t)cap * nsets, r = sizeof(uint32_t) * (4 << DG_CUT_MAX) * (size_t)nsets;
    if (tri_bytes) *tri_bytes = t;
    if (xy_bytes) *xy_bytes = x;
    if (res_bytes) *res_bytes = r;
    return t + x + r;
}

static DgScratch dg_scratch(const DelaunayScratch &h) { return DgScratch{(GTri *)h.tri, h.xy, h.res, h.cap}; }

// Launches one workgroup per set; lds = the largest delaunay_gpu_lds_bytes among them (<= 160 KB).
int launch_delaunay_gpu(const int4 *sets, int nsets, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count, size_t lds, int narrow, hipStream_t st, long long *d_level_clock) {
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
    hipLaunchKernelGGL(k_delaunay, dim3(nsets), dim3(DG_THREADS), lds, st, sets, order, xy, tri_out, tri_count, narrow, d_level_clock);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Both triangulations of every pair of a chunk, straight into the chunk's device blob.
void launch_delaunay_blob(int32_t *blob, int n_pairs, size_t lds, int sub_max, hipStream_t st) {
    static std::atomic<size_t> granted[64];
    ensure_dynamic_lds(k_delaunay_blob, lds, granted, "delaunay_gpu");
    SV_LAUNCH(K_DELAUNAY, k_delaunay_blob, dim3(2 * n_pairs), dim3(DG_THREADS), lds, st, blob, sub_max);
}

// Sets of more than sub_max vertices: 2^c subtrees per set in LDS, then the upper merges in the scratch mesh (one launch each for all
// sets; workgroups of sets that are not large, or of subtrees a set does not have, return at once).  m_max: the largest set.
int launch_delaunay_gpu_large(const int4 *sets, int nsets, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count, int m_max, int sub_max,
                              const DelaunayScratch &scratch, int narrow, hipStream_t st) {
    if (sub_max < 6 || sub_max > DG_SUB_MAX || m_max > scratch.cap || dg_cut_depth(m_max, sub_max) > DG_CUT_MAX) return -1;
    const size_t lds = delaunay_gpu_lds_bytes(sub_max, sub_max);
    static std::atomic<size_t> granted[64];
    try {
        ensure_dynamic_lds(k_dgl_subtrees, lds, granted, "delaunay_gpu");
    } catch (const std::exception &) {
        return -1;
    }
    hipLaunchKernelGGL(k_dgl_subtrees, dim3(1 << dg_cut_depth(m_max, sub_max), nsets), dim3(DG_THREADS), lds, st, sets, order, xy, tri_out, tri_count, sub_max, dg_scratch(scratch), narrow);
    hipLaunchKernelGGL(k_dgl_top, dim3(nsets), dim3(DG_THREADS), 0, st, sets, order, xy, tri_out, tri_count, sub_max, dg_scratch(scratch), narrow);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void launch_delaunay_blob_large(int32_t *blob, int n_pairs, int ns_max, int sub_max, const DelaunayScratch &scratch, hipStream_t st) {
    dg_check_cut(ns_max, sub_max, scratch);
    const size_t lds = delaunay_gpu_lds_bytes(sub_max, sub_max);
    static std::atomic<size_t> granted[64];
    ensure_dynamic_lds(k_dgl_subtrees_blob, lds, granted, "delaunay_gpu");
    SV_LAUNCH(K_DELAUNAY, k_dgl_subtrees_blob, dim3(1 << dg_cut_depth(ns_max, sub_max), 2 * n_pairs), dim3(DG_THREADS), lds, st, blob, sub_max, dg_scratch(scratch));
    SV_LAUNCH(K_DELAUNAY, k_dgl_top_blob, dim3(2 * n_pairs), dim3(DG_THREADS), 0, st, blob, sub_max, dg_scratch(scratch));
}
#endif  // DG_HOST_EMULATION

}  // namespace sv
