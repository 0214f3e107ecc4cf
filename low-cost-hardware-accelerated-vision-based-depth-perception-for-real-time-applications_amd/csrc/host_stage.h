// Host-side stages that sit between the two GPU phases of a pair: the order-dependent in-place filters of the
// support lattice and the Delaunay triangulations.  They run on the worker threads of the engine, one pair per
// thread, while other workers' kernels occupy the GPU.
#pragma once

#include <stdint.h>

#include <vector>

#include "../../include/stereo_vision_hip.h"

namespace sv {

// Support lattice -> support point list.
//   reference: serial_includes/elas/elas.cpp:152-176 (removeInconsistentSupportPoints, in place, u outer / v inner)
//              :178-233 (removeRedundantSupportPoints, vertical then horizontal, :419-420)
//              :422-433 (collect, u outer / v inner) and :235-264 (addCornerSupportPoints)
// `dcan` is [Hc][Wc] (row/col 0 hold 0 = "valid d=0", elas.cpp:387) and is modified in place.
// Writes (u,v,d) triples to `out` (capacity cap points); returns the point count (or -needed if cap is too small).
int support_filter(const sv_params &p, int16_t *dcan, int W, int H, int32_t *out, int cap);
// The transposed-lattice entry point may read (never write) up to LATTICE_PAD int16 elements past the end of T.
constexpr int LATTICE_PAD = 16;
// Same on the transposed lattice T[uc * Hc + vc] (the layout the GPU writes and the filters scan contiguously).
//
// With a team (latency mode: one pair at a time and idle cores next to the calling thread) the lattice is filtered by several threads:
// run(ctx, parts, fn, arg) calls fn(arg, part) for every part in [0, parts) - on any threads, the caller included - and returns when
// all of them have returned.  Same result, entry for entry (host_stage.cpp explains why the order-dependent filter allows it).
struct FilterTeam {
    void (*run)(void *ctx, int parts, void (*fn)(void *arg, int part), void *arg);
    void *ctx;
    int threads;  // threads the team can count on (1: not worth it)
};
struct Undecided {   // a point whose fate depends on which of the points before it were kept
    int32_t pos;     // uc * Hc + vc
    int32_t later;   // its support from itself and the points after it
};
struct FilterScratch {
    std::vector<int16_t> kept;                        // the lattice of the points kept so far
    std::vector<std::vector<Undecided>> undecided;    // per part, in scan order
};
bool support_filter_team_usable(const sv_params &p);  // AVX2 and parameters in the vector code's range
int support_filter_t(const sv_params &p, int16_t *T, int W, int H, int32_t *out, int cap, const FilterTeam *team = nullptr, FilterScratch *scratch = nullptr);
// Test hook: support_filter with a team of `threads` std::threads.
int support_filter_threads(const sv_params &p, int16_t *dcan, int W, int H, int32_t *out, int cap, int threads);

// Divide-and-conquer Delaunay triangulation with alternating cuts that reproduces, triangle for triangle and
// corner for corner, what the reference obtains from Triangle 1.6 with switches "zQB" (elas.cpp:483-484;
// common_includes/elas/triangle.cpp:5183-5924, :7449-7500).  Coordinates are integers (support points live on a
// pixel lattice), so orientation and in-circle signs are evaluated exactly in 64-bit integer arithmetic.
// The object owns its scratch memory and is reused from pair to pair (no allocation in steady state).
class Delaunay {
   public:
    // Optional helper for latency mode: run(ctx, fn, arg) makes some other thread call fn(arg) soon.  fn may also be called by
    // the triangulation itself; it does its work exactly once, and `arg` stays valid until triangulate() returns.
    struct Spawn {
        void (*run)(void *ctx, void (*fn)(void *), void *arg);
        void *ctx;
        int depth = 1;  // levels of the recursion whose right halves are handed over: 1 = two halves, 2 = four quarters
    };
    // xy: n points (x0,y0,x1,y1,...).  tri_out receives 3*count vertex indices; returns count (<= 2n), or -1 if cap
    // (in triangles) is too small.  With `spawn` the halves (quarters, ...) of the top-level cuts are built concurrently.
    int triangulate(const int32_t *xy, int n, int32_t *tri_out, int cap, const Spawn *spawn = nullptr);
    // Only the preparation (sort, duplicate scan, k-d ordering): the ids of the m surviving vertices in the order the recursion
    // consumes them; the GPU triangulation (delaunay_gpu.hip) starts from there.  Returns m (or < 0 like triangulate).
    int kd_ordered_ids(const int32_t *xy, int n, int32_t *ids_out);

   private:
    struct Tri {
        int32_t nbr[3];  // neighbour handle across edge o: (slot << 2) | orientation
        int32_t vtx[3];  // vertex ids, -1 = the vertex at infinity of a bounding ("ghost") triangle
    };
    typedef int32_t H;  // oriented-triangle handle: (slot << 2) | orientation

    struct Pt {  // sort element: the coordinates travel with the vertex id as one biased (x << 16 | y) key, so a lexicographic
        uint32_t key;  // (x, y) comparison is one unsigned compare and (y, x) order is the same key rotated by 16 bits
        int32_t id;
    };

    const int32_t *xy_ = nullptr;
    std::vector<Tri> tris_;
    std::vector<Pt> order_, sorted_;
    std::vector<uint64_t> kd_;
    int n_slots_ = 0;
    uint32_t seed_ = 1;

    int prepare(const int32_t *xy, int n);
    H make(int &cursor);
    void sort_xy(Pt *a, int n);
    void radix_sort_xy(Pt *a, int n);
    void kd_order(uint64_t *xs, uint64_t *xalt, uint64_t *ys, uint64_t *yalt, int n, int axis, Pt *out);
    void alternate_cuts(Pt *a, int m);
    void build(const Pt *a, int n, int axis, H &farleft, H &farright, int &cursor);
    void build_split(const Pt *p, int n, int axis, H &farleft, H &farright, int cursor0, int depth, const Spawn *spawn, int &cursor_end);
    void merge(H &farleft, H &innerleft, H &innerright, H &farright, int axis, int &cursor);
    uint32_t rnd(uint32_t choices);
};

}  // namespace sv
