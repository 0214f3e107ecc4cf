// The on-GPU preparation of a vertex set (csrc/delaunay_gpu.hip: dg_prepare in LDS, dg_prepare_global in scratch - bit maps, u16 rank
// prefixes, in-place partitions) executed on the CPU under ASan + UBSan: one fiber per GPU thread of the workgroup, a scheduler that
// stops a fiber at __syncthreads / __shfl_up until the others have arrived, __atomic builtins for the LDS / workgroup-scope atomics.
// The device source is compiled unchanged; the buffers have exactly the sizes the launchers request (delaunay_resident_lds_bytes,
// DgPrepScratch::words_per_set), so an index that strays past them is a heap overflow the sanitizer reports.  Against
// Delaunay::kd_ordered_ids (host_stage.cpp) - what the reference does with its quicksort and quickselects (triangle.cpp:5183-5360).
#define DG_HOST_EMULATION 1
#define DG_SIMT_EMULATION 1
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __HIP_MEMORY_SCOPE_WORKGROUP 0

#include <stdint.h>
#include <ucontext.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define SIMT_ASAN 1
#endif

// One workgroup = `n` fibers on this one OS thread (ucontext), resumed round-robin; a fiber runs until it reaches a barrier or returns.
// Deterministic, and unforgiving in the way that matters here: after a barrier fiber 0 runs ALL the way to the next barrier before
// fiber 1 has read anything - a flag that is tested after barrier N and written before barrier N + 1 is seen differently by the two
// (the return values of the workgroup's threads are compared below).
namespace simt {
struct Fiber {
    ucontext_t ctx;
    std::unique_ptr<char[]> stack;
    bool done = false;
    int wait = 0;  // 0 runnable, 1 at the workgroup barrier, 2 at its wavefront's exchange
    int gen = 0;   // ... of this generation
    int nbar = 0;  // workgroup barriers this thread has executed
};
struct Block {
    int n = 0, cur = -1, alive = 0, arrived = 0, gen = 0;
    std::vector<Fiber> f;
    std::vector<int> w_arrived, w_gen, w_alive;
    std::vector<std::array<int, 64>> xchg;
    ucontext_t sched;
    void (*body)(void *, int) = nullptr;
    void *arg = nullptr;
};
Block *blk = nullptr;
int tid = 0;
constexpr size_t STACK = 256 << 10;
int divergent_runs = 0;  // workgroups whose threads did not all execute the same number of __syncthreads (formally undefined on the device)
const void *main_bottom = nullptr;  // the scheduler's stack (the OS thread's own), as ASan reported it at the first switch
size_t main_size = 0;

inline void to_sched() {
    Fiber &me = blk->f[blk->cur];
#ifdef SIMT_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(me.done ? nullptr : &fake, main_bottom, main_size);  // (the scheduler runs on the thread's own stack)
#endif
    swapcontext(&me.ctx, &blk->sched);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

void trampoline() {
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &main_bottom, &main_size);
#endif
    Block *b = blk;
    const int me = b->cur;
    b->body(b->arg, me);
    b->f[me].done = true;
    b->alive--;
    b->w_alive[me >> 6]--;
    // a thread that leaves while others wait (a non-uniform return) must not leave them waiting for ever
    if (b->alive > 0 && b->arrived == b->alive) b->arrived = 0, b->gen++;
    if (b->w_alive[me >> 6] > 0 && b->w_arrived[me >> 6] == b->w_alive[me >> 6]) b->w_arrived[me >> 6] = 0, b->w_gen[me >> 6]++;
    to_sched();
    abort();  // never resumed
}

template <class F>
void run_block(int nthreads, F body) {
    Block b;
    b.n = b.alive = nthreads;
    b.f.resize(nthreads);
    const int nw = (nthreads + 63) / 64;
    b.w_arrived.assign(nw, 0), b.w_gen.assign(nw, 0), b.w_alive.assign(nw, 0), b.xchg.resize(nw);
    for (int t = 0; t < nthreads; t++) b.w_alive[t >> 6]++;
    b.body = [](void *a, int t) { (*static_cast<F *>(a))(t); };
    b.arg = &body;
    blk = &b;
    for (int t = 0; t < nthreads; t++) {
        Fiber &f = b.f[t];
        f.stack.reset(new char[STACK]);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.get();
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, trampoline, 0);
    }
    for (int left = nthreads; left > 0;) {
        bool progress = false;
        for (int t = 0; t < nthreads; t++) {
            Fiber &f = b.f[t];
            if (f.done) continue;
            if (f.wait == 1 && f.gen == b.gen) continue;
            if (f.wait == 2 && f.gen == b.w_gen[t >> 6]) continue;
            f.wait = 0;
            b.cur = t;
            tid = t;
            progress = true;
#ifdef SIMT_ASAN
            void *fake = nullptr;
            __sanitizer_start_switch_fiber(&fake, f.stack.get(), STACK);
#endif
            swapcontext(&b.sched, &f.ctx);
#ifdef SIMT_ASAN
            __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
            if (f.done) left--;
        }
        if (!progress) {
            printf("DEADLOCK: %d threads wait at a barrier the others never reach (divergent barrier)\n", left);
            exit(2);
        }
    }
    for (int t = 1; t < nthreads; t++)
        if (b.f[t].nbar != b.f[0].nbar) {
            if (divergent_runs++ < 4) printf("DIVERGENT BARRIERS: thread 0 executed %d __syncthreads, thread %d executed %d\n", b.f[0].nbar, t, b.f[t].nbar);
            break;
        }
    blk = nullptr;
}

inline void block_barrier() {
    Block *b = blk;
    Fiber &me = b->f[b->cur];
    me.nbar++;
    if (++b->arrived == b->alive) {
        b->arrived = 0, b->gen++;
        return;
    }
    me.wait = 1, me.gen = b->gen;
    to_sched();
}
inline void wave_barrier() {
    Block *b = blk;
    const int w = b->cur >> 6;
    Fiber &me = b->f[b->cur];
    if (++b->w_arrived[w] == b->w_alive[w]) {
        b->w_arrived[w] = 0, b->w_gen[w]++;
        return;
    }
    me.wait = 2, me.gen = b->w_gen[w];
    to_sched();
}
}  // namespace simt

struct SimtIdx {
    struct X {
        operator unsigned() const { return (unsigned)simt::tid; }
    } x;
};
static SimtIdx threadIdx;
static inline void __syncthreads() {
    simt::block_barrier();
    simt::tid = simt::blk->cur;
}
static inline int __shfl_up(int v, int off, int) {  // all lanes of the wavefront call it (converged code)
    const int me = simt::blk->cur, w = me >> 6, lane = me & 63;
    simt::blk->xchg[w][lane] = v;
    simt::wave_barrier();
    const int r = lane >= off ? simt::blk->xchg[w][lane - off] : v;
    simt::wave_barrier();
    simt::tid = me;
    return r;
}
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
using std::max;
using std::min;
template <class T> static inline T __hip_atomic_fetch_or(T *p, T v, int, int) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T __hip_atomic_fetch_add(T *p, T v, int, int) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T __hip_atomic_exchange(T *p, T v, int, int) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T __hip_atomic_load(const T *p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <class T> static inline void __hip_atomic_store(T *p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
template <class T> static inline T __hip_atomic_fetch_min(T *p, T v, int, int) {
    T cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
    return cur;
}
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicMin(int *p, int v) { return __hip_atomic_fetch_min(p, v, 0, 0); }

#include "delaunay_gpu.hip"
#include "host_stage.h"

using namespace sv::dg;

struct Case {
    int W, H, step, disp_max;
    std::vector<int32_t> sup;  // (u, v, d) triples
};

// support points of a W x H image: distinct lattice cells, the four image corners on request (the last line is a row of its own when
// it is no lattice row).  Disparities grow with u slower than u itself, so u - d is distinct within a line as well - real maps are
// smooth; coincident right-image vertices are a case of their own below.
static Case make_case(std::mt19937 &rng, int W, int H, int step, int disp_max, int n, bool corners) {
    Case c{W, H, step, disp_max, {}};
    const int Wc = (W - 1) / step, Hc = (H - 1) / step, dm = disp_max - 2;
    std::vector<int> cells((size_t)Wc * Hc);
    for (size_t i = 0; i < cells.size(); i++) cells[i] = (int)i;
    std::shuffle(cells.begin(), cells.end(), rng);
    n = std::min<int>(n, (int)cells.size());
    auto disp = [&](int u) { return (int)((long)u * dm / W); };
    for (int i = 0; i < n; i++) {
        const int uc = cells[i] % Wc + 1, vc = cells[i] / Wc + 1;
        c.sup.insert(c.sup.end(), {uc * step, vc * step, disp(uc * step)});
    }
    if (corners)
        for (int k = 0; k < 4; k++) {
            const int u = (k & 1) ? W - 1 : 0, v = (k & 2) ? H - 1 : 0;
            bool taken = false;  // (a corner that is a lattice point as well: W - 1 and H - 1 multiples of the step)
            for (size_t q = 0; q + 2 < c.sup.size(); q += 3) taken = taken || (c.sup[q] == u && c.sup[q + 1] == v);
            if (taken) continue;
            int d = disp(u);
            for (bool clash = true; clash && d > 0;) {  // (the last line can be a lattice row: keep u - d off its points)
                clash = false;
                for (size_t q = 0; q + 2 < c.sup.size(); q += 3)
                    if (c.sup[q + 1] == v && c.sup[q] - c.sup[q + 2] == u - d) clash = true;
                if (clash) d--;
            }
            c.sup.insert(c.sup.end(), {u, v, d});
        }
    return c;
}

static int bad = 0;
static void fail(const char *what, int it, int side, int a, int b) {
    if (++bad < 8) printf("MISMATCH %s: case %d side %d: %d vs %d\n", what, it, side, a, b);
}

// the host's answer: m and the vertices' coordinates in k-d order (ids of coincident twins may differ, their coordinates may not)
static int host_order(const Case &c, int side, std::vector<int32_t> &xy, std::vector<int32_t> &ids) {
    const int n = (int)c.sup.size() / 3;
    xy.resize(2 * (size_t)n);
    ids.assign(n, 0);
    for (int i = 0; i < n; i++) xy[2 * i] = c.sup[3 * i] - (side ? c.sup[3 * i + 2] : 0), xy[2 * i + 1] = c.sup[3 * i + 1];
    sv::Delaunay dl;
    return dl.kd_ordered_ids(xy.data(), n, ids.data());
}

// dg_prepare on DG_THREADS CPU threads; expect: the host's m, or -1 for a set the kernel must hand back
static void check_lds(const Case &c, int it, int side, int expect_m) {
    const int n = (int)c.sup.size() / 3;
    const DgPrep pp = sv::dg_prep_dims(c.W, c.H, c.step, c.disp_max);
    const size_t bytes = sv::delaunay_resident_lds_bytes(c.W, c.H, c.step, c.disp_max, n < 3 ? 3 : n);
    if (bytes > 160 * 1024) return;  // (the launcher would not take the LDS path)
    std::vector<uint32_t> lds((bytes - 16) / 4);  // exactly what the kernel may touch (16: the request's alignment slack)
    std::vector<int32_t> dsp(n);
    for (int i = 0; i < n; i++) dsp[i] = c.sup[3 * i + 2];
    const DgLds L = dg_carve(lds.data(), n, n);
    for (int i = 0; i < n; i++) L.set_vertex(i, c.sup[3 * i] - (side ? dsp[i] : 0), c.sup[3 * i + 1]);
    std::vector<int> ret(DG_THREADS, -7);
    simt::run_block(DG_THREADS, [&](int t) { ret[t] = dg_prepare(L, n, pp, dsp.data(), 1); });
    for (int t = 1; t < DG_THREADS; t++)
        if (ret[t] != ret[0]) fail("return value not uniform", it, side, ret[0], ret[t]);
    if (ret[0] != expect_m) return fail("m (LDS)", it, side, expect_m, ret[0]);
    if (expect_m < 0) return;
    std::vector<int32_t> xy, ids;
    host_order(c, side, xy, ids);
    for (int i = 0; i < expect_m; i++) {
        const int g = L.ord[i], h = ids[i];
        if (g >= n || xy[2 * g] != xy[2 * h] || xy[2 * g + 1] != xy[2 * h + 1]) return fail("order (LDS)", it, side, h, g);
    }
}

static void check_global(const Case &c, int it, int side, int cap, int expect_m) {
    const int n = (int)c.sup.size() / 3;
    const DgPrep pp = sv::dg_prep_dims(c.W, c.H, c.step, c.disp_max);
    const DgPrepScratch sc{nullptr, cap, pp.bm_words};
    std::vector<uint32_t> scratch(sc.words_per_set());
    std::vector<int32_t> ord((size_t)n + 1, -9);
    std::vector<int> ret(DGP_THREADS, -7);
    simt::run_block(DGP_THREADS, [&](int t) { ret[t] = dg_prepare_global(c.sup.data(), n, side, pp, scratch.data(), cap, pp.bm_words, ord.data()); });
    for (int t = 1; t < DGP_THREADS; t++)
        if (ret[t] != ret[0]) fail("return value not uniform (global)", it, side, ret[0], ret[t]);
    if (ret[0] != expect_m || ord[0] != expect_m) return fail("m (global)", it, side, expect_m, ret[0]);
    if (expect_m < 0) return;
    std::vector<int32_t> xy, ids;
    host_order(c, side, xy, ids);
    for (int i = 0; i < expect_m; i++) {
        const int g = ord[1 + i], h = ids[i];
        if (g < 0 || g >= n || xy[2 * g] != xy[2 * h] || xy[2 * g + 1] != xy[2 * h + 1]) return fail("order (global)", it, side, h, g);
    }
}

int main(int argc, char **argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;  // 0: a quick pass
    std::mt19937 rng(77);
    int it = 0, handed_back = 0;
    std::vector<int32_t> xy, ids;
    // ---- LDS path: KITTI-sized and small images, odd sizes (the last line a row of its own or not), both sides
    const int sizes[][4] = {{1242, 375, 5, 127}, {1242, 375, 5, 255}, {320, 120, 5, 63}, {641, 241, 5, 63}, {621, 187, 10, 63}, {64, 48, 5, 15}, {1242, 376, 5, 127}};
    for (int rep = 0; rep < (scale ? 3 : 1); rep++)
        for (const auto &sz : sizes) {
            const int lat = ((sz[0] - 1) / sz[2]) * ((sz[1] - 1) / sz[2]);
            const int n = rep == 0 ? std::min(lat, 2100) : 3 + (int)(rng() % (unsigned)std::min(lat, 3900));
            Case c = make_case(rng, sz[0], sz[1], sz[2], sz[3], n, rep != 1);
            for (int side = 0; side < 2; side++) check_lds(c, it, side, host_order(c, side, xy, ids));
            it++;
        }
    {  // m = DG_PREP_MAX exactly, and one more (handed back)
        Case c = make_case(rng, 1242, 375, 5, 63, DG_PREP_MAX, false);
        check_lds(c, it++, 0, host_order(c, 0, xy, ids));
        c.sup.insert(c.sup.end(), {0, 0, 0});
        check_lds(c, it++, 0, -1), handed_back++;
    }
    for (int k = 0; k < (scale ? 6 : 2); k++) {  // coincident vertices
        Case c = make_case(rng, 1242, 375, 5, 127, 1500 + 100 * k, true);
        const int n0 = (int)c.sup.size() / 3;
        const int groups = k == 0 ? 1 : (k == 1 ? DG_DUP_MAX : (k == 2 ? DG_DUP_MAX + 1 : 1 + (int)(rng() % 6)));
        for (int g = 0; g < groups; g++) {  // the same (u, v, d) triple again: the same support point twice (elas.cpp:258-259 does that to a corner)
            const int src = (int)(rng() % (unsigned)n0);
            c.sup.insert(c.sup.end(), {c.sup[3 * src], c.sup[3 * src + 1], c.sup[3 * src + 2]});
        }
        if (k == 3) c.sup[c.sup.size() - 1] += 1;  // ... with another disparity: not interchangeable, the host's business
        const bool back = k == 2 || k == 3;
        handed_back += back;
        // (left side: equal (u, v); right side: u - d equal as well since d is equal - except k == 3, where the twin moves to another cell or not)
        check_lds(c, it, 0, back ? -1 : host_order(c, 0, xy, ids));
        if (k != 3) check_lds(c, it, 1, back ? -1 : host_order(c, 1, xy, ids));
        it++;
    }
    {  // a vertex outside the bit maps (x beyond W - 1 + disp_max): handed back
        Case c = make_case(rng, 320, 120, 5, 63, 300, true);
        c.sup.insert(c.sup.end(), {320 + 63 + 1, 5, 0});
        check_lds(c, it++, 0, -1), handed_back++;
    }
    // ---- global-memory path (4K lattices): the same properties with 1 024 threads, fewer cases
    for (int k = 0; k < (scale ? 5 : 2); k++) {
        const int n = k == 0 ? 21000 : 4001 + (int)(rng() % 26000u);
        Case c = make_case(rng, 3840, 2160, 5, 191, n, k % 2 == 0);
        if (k == 1) {  // coincident corner
            c.sup.insert(c.sup.end(), {c.sup[0], c.sup[1], c.sup[2]});
        }
        const int cap = (int)c.sup.size() / 3 + (k == 2 ? 0 : 37);
        for (int side = 0; side < 2; side++) check_global(c, it, side, cap, host_order(c, side, xy, ids));
        it++;
    }
    {  // more vertices than the scratch holds; a coincident pair with different disparities
        Case c = make_case(rng, 3840, 2160, 5, 191, 5000, true);
        check_global(c, it++, 0, 4999, -1), handed_back++;
        c.sup.insert(c.sup.end(), {c.sup[0], c.sup[1], c.sup[2] + 1});
        check_global(c, it++, 0, 6000, -1), handed_back++;
    }
    printf("preparation emulated: %d cases, %d handed back as expected\n", it, handed_back);
    bad += simt::divergent_runs;
    printf("dg-prepare emulation done, mismatches: %d\n", bad);
    return bad != 0;
}
