// Host-side stages between the two GPU phases (see host_stage.h).  Built with -ffp-contract=off like everything else,
// although this file is integer-only.
#include "host_stage.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace sv {

// ------------------------------------------------------------------------------------------------------------
// support lattice filters
// ------------------------------------------------------------------------------------------------------------

static int lattice_step(const sv_params &p) {  // elas.cpp:376-378: an even step at half resolution
    return p.subsampling ? p.candidate_stepsize + p.candidate_stepsize % 2 : p.candidate_stepsize;
}

static void lattice_dims(const sv_params &p, int W, int H, int &Wc, int &Hc) {  // elas.cpp:376-386
    const int step = lattice_step(p);
    Wc = (W + step - 1) / step;
    Hc = (H + step - 1) / step;
}

// All three filters scan the lattice u outer / v inner (elas.cpp:154-155, 196-197), so the lattice is kept TRANSPOSED
// here, T[uc * Hc + vc]: the inner loop then walks contiguous memory (the GPU writes it in this layout).

// elas.cpp:152-176.  In place and order dependent: a point invalidated earlier in the scan no longer supports later
// points.  Counting stops as soon as incon_min_support is reached (the reference only tests `<`).
static void drop_inconsistent_scalar(const sv_params &p, int16_t *T, int Wc, int Hc) {
    const int win = p.incon_window_size, thr = p.incon_threshold, need = p.incon_min_support;
    for (int uc = 0; uc < Wc; uc++) {
        const int u_lo = std::max(uc - win, 0), u_hi = std::min(uc + win, Wc - 1);
        int16_t *col = T + (size_t)uc * Hc;
        for (int vc = 0; vc < Hc; vc++) {
            const int d = col[vc];
            if (d < 0) continue;
            const int v_lo = std::max(vc - win, 0), v_hi = std::min(vc + win, Hc - 1);
            int support = 0;
            for (int u2 = u_lo; u2 <= u_hi && support < need; u2++) {
                const int16_t *c2 = T + (size_t)u2 * Hc;
                for (int v2 = v_lo; v2 <= v_hi; v2++) {
                    const int d2 = c2[v2];
                    support += (d2 >= 0) & (abs(d - d2) <= thr);
                }
            }
            if (support < need) col[vc] = -1;
        }
    }
}

#if defined(__x86_64__)
// Valid entries (>= 0) of col[v0 .. v0+15] as a bit mask, 2 bits per int16 lane (movemask_epi8), rows >= Hc cleared.
// An entry can only be invalidated when it is visited itself, so the mask stays right for the entries not yet visited.
__attribute__((target("avx2"))) static inline uint32_t valid_lanes(const int16_t *col, int v0, int Hc) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col + v0));
    uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi16(x, _mm256_set1_epi16(-1)));
    const int rows = Hc - v0;
    if (rows < 16) m &= (1u << (2 * rows)) - 1u;
    return m & 0x55555555u;  // one bit per lane
}

// Same scan with one 16-lane int16 compare per window column (the window's <= 11 rows are contiguous in the transposed
// lattice).  Reads up to 15 elements past a column's window: the caller's buffer is padded accordingly (LATTICE_PAD).
// Invalid entries are skipped 16 at a time through their validity mask instead of one mispredicted branch each.
__attribute__((target("avx2"))) static void drop_inconsistent_avx2(const sv_params &p, int16_t *T, int Wc, int Hc) {
    const int win = p.incon_window_size, need = p.incon_min_support;
    const __m256i vthr = _mm256_set1_epi16((short)p.incon_threshold), vneg1 = _mm256_set1_epi16(-1);
    for (int uc = 0; uc < Wc; uc++) {
        const int u_lo = std::max(uc - win, 0), u_hi = std::min(uc + win, Wc - 1);
        int16_t *col = T + (size_t)uc * Hc;
        for (int v0 = 0; v0 < Hc; v0 += 16)
            for (uint32_t vm = valid_lanes(col, v0, Hc); vm; vm &= vm - 1) {
                const int vc = v0 + (__builtin_ctz(vm) >> 1);
                const int d = col[vc];
                const int v_lo = std::max(vc - win, 0), v_hi = std::min(vc + win, Hc - 1);
                const uint32_t lanes = (1u << (2 * (v_hi - v_lo + 1))) - 1u;  // movemask gives 2 bits per int16 lane
                const __m256i vd = _mm256_set1_epi16((short)d);
                int support = 0;
                for (int u2 = u_lo; u2 <= u_hi && support < need; u2++) {
                    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(T + (size_t)u2 * Hc + v_lo));
                    const __m256i diff = _mm256_abs_epi16(_mm256_sub_epi16(x, vd));
                    const __m256i ok = _mm256_andnot_si256(_mm256_cmpgt_epi16(diff, vthr), _mm256_cmpgt_epi16(x, vneg1));
                    support += __builtin_popcount((uint32_t)_mm256_movemask_epi8(ok) & lanes) >> 1;
                }
                if (support < need) col[vc] = -1;
            }
    }
}
#endif

static void drop_inconsistent(const sv_params &p, int16_t *T, int Wc, int Hc) {
#if defined(__x86_64__)
    // disparities are < 1024 and the window has at most 11 rows: the int16 arithmetic cannot overflow
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2 && p.incon_window_size <= 7 && p.incon_threshold >= 0 && p.incon_threshold < 16384) {
        drop_inconsistent_avx2(p, T, Wc, Hc);
        return;
    }
#endif
    drop_inconsistent_scalar(p, T, Wc, Hc);
}

// elas.cpp:178-233 with redun_max_dist = 5, redun_threshold = 1 (:419-420); in place.  A point is dropped when, in BOTH
// directions along the axis, some valid point with |dd| <= thr lies within max_dist steps ("first found" == "any found").
// Branch-free per direction: the data-dependent early exits of the reference mispredict on almost every point.
static void drop_redundant_scalar(int16_t *T, int Wc, int Hc, int max_dist, int thr, bool vertical) {
    const int stride = vertical ? 1 : Hc;
    for (int uc = 0; uc < Wc; uc++)
        for (int vc = 0; vc < Hc; vc++) {
            int16_t *q = T + (size_t)uc * Hc + vc;
            const int d = *q;
            if (d < 0) continue;
            const int pos = vertical ? vc : uc, len = vertical ? Hc : Wc;
            const int n_lo = std::min(max_dist, pos), n_hi = std::min(max_dist, len - 1 - pos);  // steps available before the border
            int found_lo = 0, found_hi = 0;
            for (int j = 1; j <= n_lo; j++) {
                const int d2 = q[-j * stride];
                found_lo |= (d2 >= 0) & (abs(d - d2) <= thr);
            }
            for (int j = 1; j <= n_hi; j++) {
                const int d2 = q[j * stride];
                found_hi |= (d2 >= 0) & (abs(d - d2) <= thr);
            }
            if (found_lo & found_hi) *q = -1;
        }
}

#if defined(__x86_64__)
// Pass along v (the contiguous axis of the transposed lattice): the scan is sequential inside a column, so every point is
// still visited in order, but its +-max_dist neighbourhood is one 16-lane window load: lanes 0..max_dist-1 are the points
// below, lane max_dist the point itself, the next max_dist lanes the points above.  Points closer than max_dist to a column
// end take the scalar path.  Requires max_dist <= 7.
__attribute__((target("avx2"))) static void drop_redundant_v_avx2(int16_t *T, int uc0, int uc1, int Hc, int max_dist, int thr) {  // columns [uc0, uc1)
    const __m256i vthr = _mm256_set1_epi16((short)thr), vneg1 = _mm256_set1_epi16(-1);
    const uint32_t lo_bits = (1u << (2 * max_dist)) - 1u;                              // movemask: 2 bits per int16 lane
    const uint32_t hi_bits = ((1u << (2 * max_dist)) - 1u) << (2 * (max_dist + 1));
    // A dropped point is NOT written while its column is being scanned: a 2-byte store followed by 32-byte loads that cover it (the
    // windows of the next points) cannot be forwarded from the store buffer and stalls every one of them - and this filter drops five
    // points of six (a 4K lattice: 370 - 510 us for this pass, more than the whole parallel classification of the first filter).  The rows
    // dropped so far travel in a register instead (hist: bit j = row vc - 1 - j was dropped), the stores follow when the column is done.
    uint32_t lane_clear[128];  // movemask bits of the window lanes 0 .. max_dist - 1 that hist's low bits take out (lane i = row vc - max_dist + i)
    for (int h = 0; h < (1 << max_dist); h++) {
        uint32_t m = 0;
        for (int jj = 0; jj < max_dist; jj++)
            if ((h >> jj) & 1) m |= 3u << (2 * (max_dist - 1 - jj));
        lane_clear[h] = m;
    }
    static thread_local std::vector<int16_t> drops;
    if ((int)drops.size() < Hc) drops.resize(Hc);
    for (int uc = uc0; uc < uc1; uc++) {
        int16_t *col = T + (size_t)uc * Hc;
        uint64_t hist = 0;
        int hist_row = 0, ndrop = 0;  // bit j of hist: row hist_row - 1 - j
        for (int v0 = 0; v0 < Hc; v0 += 16) {
            uint32_t vm;
            if (uc == uc1 - 1 && Hc - v0 < 16) {  // the range's last column, its last rows: not a 16-lane load that runs into the next column (another thread's, perhaps)
                alignas(32) int16_t tmp[16];
                for (int i = 0; i < 16; i++) tmp[i] = i < Hc - v0 ? col[v0 + i] : (int16_t)-1;
                vm = valid_lanes(tmp, 0, 16);
            } else {
                vm = valid_lanes(col, v0, Hc);
            }
            for (; vm; vm &= vm - 1) {
                const int vc = v0 + (__builtin_ctz(vm) >> 1);
                const int d = col[vc];
                const int sh = vc - hist_row;
                hist = sh >= 64 ? 0 : hist << sh;
                hist_row = vc;
                bool drop;
                // near a column end; in the last column of the range also wherever the 16-lane window would reach into the next column, which
                // may be another thread's (its lanes would be ignored, but the read itself would race with that thread's writes)
                if (vc < max_dist || vc + max_dist >= Hc || (uc == uc1 - 1 && vc - max_dist + 16 > Hc)) {
                    const int n_lo = std::min(max_dist, vc), n_hi = std::min(max_dist, Hc - 1 - vc);
                    int found_lo = 0, found_hi = 0;
                    for (int jj = 1; jj <= n_lo; jj++) found_lo |= (col[vc - jj] >= 0) & (abs(d - col[vc - jj]) <= thr) & (int)(~(hist >> (jj - 1)) & 1u);
                    for (int jj = 1; jj <= n_hi; jj++) found_hi |= (col[vc + jj] >= 0) & (abs(d - col[vc + jj]) <= thr);
                    drop = (found_lo & found_hi) != 0;
                } else {
                    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col + vc - max_dist));
                    const __m256i diff = _mm256_abs_epi16(_mm256_sub_epi16(x, _mm256_set1_epi16((short)d)));
                    const __m256i ok = _mm256_andnot_si256(_mm256_cmpgt_epi16(diff, vthr), _mm256_cmpgt_epi16(x, vneg1));
                    const uint32_t m = (uint32_t)_mm256_movemask_epi8(ok) & ~lane_clear[hist & ((1u << max_dist) - 1u)];
                    drop = (m & lo_bits) && (m & hi_bits);
                }
                hist = (hist << 1) | (drop ? 1u : 0u);
                hist_row = vc + 1;
                if (drop) drops[ndrop++] = (int16_t)vc;
            }
        }
        for (int i = 0; i < ndrop; i++) col[drops[i]] = -1;
    }
}

// Pass along u: a point only looks at points of its own row v, so the 16 rows of a vector are independent of each other and
// the column loop (u ascending) already is the reference's visiting order for each of them.
__attribute__((target("avx2"))) static void drop_redundant_u_avx2(int16_t *T, int Wc, int Hc, int max_dist, int thr, int vb0 = 0, int vb1 = 1 << 30) {  // rows [vb0, vb1), vb0 a multiple of 16
    const __m256i vthr = _mm256_set1_epi16((short)thr), vneg1 = _mm256_set1_epi16(-1);
    alignas(32) int16_t lane_id[16];
    for (int i = 0; i < 16; i++) lane_id[i] = (int16_t)i;
    const __m256i vlane = _mm256_load_si256(reinterpret_cast<const __m256i *>(lane_id));
    for (int uc = 0; uc < Wc; uc++) {
        const int n_lo = std::min(max_dist, uc), n_hi = std::min(max_dist, Wc - 1 - uc);
        int16_t *col = T + (size_t)uc * Hc;
        for (int v0 = vb0; v0 < std::min(Hc, vb1); v0 += 16) {
            const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col + v0));
            const __m256i valid = _mm256_cmpgt_epi16(x, vneg1);
            __m256i flo = _mm256_setzero_si256(), fhi = _mm256_setzero_si256();
            for (int j = 1; j <= n_lo; j++) {
                const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col - (size_t)j * Hc + v0));
                const __m256i near = _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(x, y)), vthr), _mm256_cmpgt_epi16(y, vneg1));
                flo = _mm256_or_si256(flo, near);
            }
            for (int j = 1; j <= n_hi; j++) {
                const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col + (size_t)j * Hc + v0));
                const __m256i near = _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(x, y)), vthr), _mm256_cmpgt_epi16(y, vneg1));
                fhi = _mm256_or_si256(fhi, near);
            }
            // rows beyond the column's end belong to the next column: leave them alone
            const __m256i inside = _mm256_cmpgt_epi16(_mm256_set1_epi16((short)std::min(Hc - v0, 16)), vlane);
            const __m256i drop = _mm256_and_si256(_mm256_and_si256(valid, inside), _mm256_and_si256(flo, fhi));
            const __m256i res = _mm256_blendv_epi8(x, vneg1, drop);
            if (Hc - v0 >= 16) {
                _mm256_storeu_si256(reinterpret_cast<__m256i *>(col + v0), res);
            } else {  // the column's last rows only: the entries behind them are the next column's first rows, which another thread of a team may be rewriting
                alignas(32) int16_t tmp[16];
                _mm256_store_si256(reinterpret_cast<__m256i *>(tmp), res);
                memcpy(col + v0, tmp, sizeof(int16_t) * (size_t)(Hc - v0));
            }
        }
    }
}
#endif

static void drop_redundant(int16_t *T, int Wc, int Hc, int max_dist, int thr, bool vertical) {
#if defined(__x86_64__)
    // 16-lane loads may run up to 15 elements past the last column: the caller's buffer is padded (LATTICE_PAD)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2 && max_dist >= 1 && max_dist <= 7 && thr >= 0 && thr < 16384) {
        if (vertical)
            drop_redundant_v_avx2(T, 0, Wc, Hc, max_dist, thr);
        else
            drop_redundant_u_avx2(T, Wc, Hc, max_dist, thr);
        return;
    }
#endif
    drop_redundant_scalar(T, Wc, Hc, max_dist, thr, vertical);
}

// elas.cpp:422-433: the surviving lattice points as (u, v, d) triples, u outer / v inner, row and column 0 excluded.
// Returns the number of points found (only the first `cap` are stored).
static int collect_points_scalar(const int16_t *T, int Wc, int Hc, int step, int32_t *out, int cap) {
    int n = 0;
    for (int uc = 1; uc < Wc; uc++)
        for (int vc = 1; vc < Hc; vc++) {
            const int d = T[(size_t)uc * Hc + vc];
            if (d < 0) continue;
            if (n < cap) {
                out[3 * n] = uc * step;
                out[3 * n + 1] = vc * step;
                out[3 * n + 2] = d;
            }
            n++;
        }
    return n;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static int collect_points_avx2(const int16_t *T, int Wc, int Hc, int step, int32_t *out, int cap) {
    int n = 0;
    for (int uc = 1; uc < Wc; uc++) {
        const int16_t *col = T + (size_t)uc * Hc;
        for (int v0 = 0; v0 < Hc; v0 += 16) {
            uint32_t vm = valid_lanes(col, v0, Hc);
            if (v0 == 0) vm &= ~1u;  // row 0 is not part of the lattice proper (elas.cpp:394)
            for (; vm; vm &= vm - 1) {
                const int vc = v0 + (__builtin_ctz(vm) >> 1);
                if (n < cap) {
                    out[3 * n] = uc * step;
                    out[3 * n + 1] = vc * step;
                    out[3 * n + 2] = col[vc];
                }
                n++;
            }
        }
    }
    return n;
}
#endif

static int collect_points(const int16_t *T, int Wc, int Hc, int step, int32_t *out, int cap) {
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) return collect_points_avx2(T, Wc, Hc, step, out, cap);
#endif
    return collect_points_scalar(T, Wc, Hc, step, out, cap);
}

#if defined(__x86_64__)
// ---- the same filters with the lattice shared between the threads of a team (latency mode: one pair, several idle cores) ----
//
// removeInconsistentSupportPoints is order dependent, but only one way: when a point is visited, the points AFTER it in the scan
// (u outer, v inner) still have their original values and the points BEFORE it have their final ones.  So
//     support(p) = L(p) + E(p),   L = similar valid points of the window at or after p (itself included) - known from the input alone,
//                                 E = similar points of the window before p that were KEPT,  0 <= E <= Emax = those valid in the input.
// Pass 1 (any order, any number of threads, reads the input only): L >= need: kept for certain; L + Emax < need: dropped for certain;
// otherwise undecided (with its L).  Pass 2 (one thread, scan order, undecided points only): E counted in the lattice of kept points.
// On a KITTI lattice 6 400 of 18 675 entries are valid and a few hundred stay undecided.
// similar valid entries among the `rows` entries from `col` (bit pairs of movemask; rows <= 16)
__attribute__((target("avx2"))) static inline uint32_t similar_lanes(const int16_t *col, __m256i vd, __m256i vthr, __m256i vneg1, uint32_t lanes) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col));
    const __m256i diff = _mm256_abs_epi16(_mm256_sub_epi16(x, vd));
    const __m256i ok = _mm256_andnot_si256(_mm256_cmpgt_epi16(diff, vthr), _mm256_cmpgt_epi16(x, vneg1));
    return (uint32_t)_mm256_movemask_epi8(ok) & lanes;
}

// pass 1 for the columns [uc0, uc1): K[pos] = d for points kept for certain, -1 for everything else; undecided points appended in scan order
__attribute__((target("avx2"))) static void classify_inconsistent_avx2(const sv_params &p, const int16_t *T, int16_t *K, int Wc, int Hc, int uc0, int uc1, std::vector<Undecided> &und) {
    const int win = p.incon_window_size, need = p.incon_min_support;
    const __m256i vthr = _mm256_set1_epi16((short)p.incon_threshold), vneg1 = _mm256_set1_epi16(-1);
    for (int uc = uc0; uc < uc1; uc++) {
        const int u_lo = std::max(uc - win, 0), u_hi = std::min(uc + win, Wc - 1);
        const int16_t *col = T + (size_t)uc * Hc;
        int16_t *kcol = K + (size_t)uc * Hc;
        std::fill_n(kcol, Hc, (int16_t)-1);  // (exactly this column: the next one may belong to another thread)
        for (int v0 = 0; v0 < Hc; v0 += 16) {
            for (uint32_t vm = valid_lanes(col, v0, Hc); vm; vm &= vm - 1) {
                const int vc = v0 + (__builtin_ctz(vm) >> 1);
                const int d = col[vc];
                const int v_lo = std::max(vc - win, 0), v_hi = std::min(vc + win, Hc - 1);
                const uint32_t lanes = (uint32_t)((1ull << (2 * (v_hi - v_lo + 1))) - 1ull);
                const uint32_t before = (1u << (2 * (vc - v_lo))) - 1u;  // the rows of a column that come before row vc
                const __m256i vd = _mm256_set1_epi16((short)d);
                const uint32_t own = similar_lanes(col + v_lo, vd, vthr, vneg1, lanes);
                int later = __builtin_popcount(own & ~before) >> 1;  // (the point itself is one of them)
                for (int u2 = uc + 1; u2 <= u_hi && later < need; u2++) later += __builtin_popcount(similar_lanes(T + (size_t)u2 * Hc + v_lo, vd, vthr, vneg1, lanes)) >> 1;
                if (later >= need) {
                    kcol[vc] = (int16_t)d;
                    continue;
                }
                int sup = later + (__builtin_popcount(own & before) >> 1);
                for (int u2 = uc - 1; u2 >= u_lo && sup < need; u2--) sup += __builtin_popcount(similar_lanes(T + (size_t)u2 * Hc + v_lo, vd, vthr, vneg1, lanes)) >> 1;
                if (sup >= need) und.push_back(Undecided{uc * Hc + vc, later});
            }
        }
    }
}

// pass 2: the undecided points in scan order against the kept points before them
__attribute__((target("avx2"))) static void resolve_inconsistent_avx2(const sv_params &p, const int16_t *T, int16_t *K, int Wc, int Hc, const Undecided *und, size_t n) {
    const int win = p.incon_window_size, need = p.incon_min_support;
    const __m256i vthr = _mm256_set1_epi16((short)p.incon_threshold), vneg1 = _mm256_set1_epi16(-1);
    for (size_t i = 0; i < n; i++) {
        const int uc = und[i].pos / Hc, vc = und[i].pos - uc * Hc;
        const int d = T[und[i].pos];
        const int u_lo = std::max(uc - win, 0);
        const int v_lo = std::max(vc - win, 0), v_hi = std::min(vc + win, Hc - 1);
        const uint32_t lanes = (uint32_t)((1ull << (2 * (v_hi - v_lo + 1))) - 1ull);
        const uint32_t before = (1u << (2 * (vc - v_lo))) - 1u;
        const __m256i vd = _mm256_set1_epi16((short)d);
        int sup = und[i].later + (__builtin_popcount(similar_lanes(K + (size_t)uc * Hc + v_lo, vd, vthr, vneg1, lanes & before)) >> 1);
        for (int u2 = uc - 1; u2 >= u_lo && sup < need; u2--) sup += __builtin_popcount(similar_lanes(K + (size_t)u2 * Hc + v_lo, vd, vthr, vneg1, lanes)) >> 1;
        if (sup >= need) K[und[i].pos] = (int16_t)d;
    }
}

// The pass along u for the rows [v0, v0 + 16) by one thread of a team, on a private copy of those rows ([Wc][16], contiguous): in the
// shared lattice the row blocks of a column lie side by side in the same cache lines, and five threads rewriting them in place spend
// their time passing those lines around (measured: 11 us for the pass on five threads, 8 on one).  Only changed entries go back.
__attribute__((target("avx2"))) static void drop_redundant_u_block_avx2(int16_t *K, int Wc, int Hc, int max_dist, int thr, int v0, std::vector<int16_t> &blk) {
    const int rows = std::min(Hc - v0, 16);
    if (rows <= 0) return;
    if (blk.size() < (size_t)Wc * 16) blk.resize((size_t)Wc * 16);
    int16_t *B = blk.data();
    if (rows == 16) {
        for (int uc = 0; uc < Wc; uc++)
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(B + (size_t)uc * 16), _mm256_loadu_si256(reinterpret_cast<const __m256i *>(K + (size_t)uc * Hc + v0)));
    } else {  // the column's last rows: exactly those (the entries behind them are the next column's first rows - another thread's)
        for (int uc = 0; uc < Wc; uc++) {
            int16_t *col = B + (size_t)uc * 16;
            for (int i = rows; i < 16; i++) col[i] = -1;
            memcpy(col, K + (size_t)uc * Hc + v0, sizeof(int16_t) * (size_t)rows);
        }
    }
    const __m256i vthr = _mm256_set1_epi16((short)thr), vneg1 = _mm256_set1_epi16(-1);
    alignas(32) int16_t lane_id[16];
    for (int i = 0; i < 16; i++) lane_id[i] = (int16_t)i;
    const __m256i inside = _mm256_cmpgt_epi16(_mm256_set1_epi16((short)rows), _mm256_load_si256(reinterpret_cast<const __m256i *>(lane_id)));
    for (int uc = 0; uc < Wc; uc++) {
        const int n_lo = std::min(max_dist, uc), n_hi = std::min(max_dist, Wc - 1 - uc);
        int16_t *col = B + (size_t)uc * 16;
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col));
        const __m256i valid = _mm256_and_si256(_mm256_cmpgt_epi16(x, vneg1), inside);
        if (_mm256_testz_si256(valid, valid)) continue;
        __m256i flo = _mm256_setzero_si256(), fhi = _mm256_setzero_si256();
        for (int j = 1; j <= n_lo; j++) {
            const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col - (size_t)j * 16));
            flo = _mm256_or_si256(flo, _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(x, y)), vthr), _mm256_cmpgt_epi16(y, vneg1)));
        }
        for (int j = 1; j <= n_hi; j++) {
            const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(col + (size_t)j * 16));
            fhi = _mm256_or_si256(fhi, _mm256_andnot_si256(_mm256_cmpgt_epi16(_mm256_abs_epi16(_mm256_sub_epi16(x, y)), vthr), _mm256_cmpgt_epi16(y, vneg1)));
        }
        const __m256i drop = _mm256_and_si256(valid, _mm256_and_si256(flo, fhi));
        if (_mm256_testz_si256(drop, drop)) continue;
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(col), _mm256_blendv_epi8(x, vneg1, drop));
        memcpy(K + (size_t)uc * Hc + v0, col, sizeof(int16_t) * (size_t)rows);  // (this thread's rows of the column: nobody else reads or writes them)
    }
}

namespace {
struct TeamJob {
    const sv_params *p;
    const int16_t *T;
    int16_t *K;
    int Wc, Hc, parts;
    std::vector<Undecided> *und;  // [parts]
    int stage;
};
}  // namespace

__attribute__((target("avx2"))) static void team_piece(void *arg, int part) {
    const TeamJob &j = *static_cast<const TeamJob *>(arg);
    const int c0 = (int)((long)j.Wc * part / j.parts), c1 = (int)((long)j.Wc * (part + 1) / j.parts);
    if (j.stage == 0) {
        j.und[part].clear();
        classify_inconsistent_avx2(*j.p, j.T, j.K, j.Wc, j.Hc, c0, c1, j.und[part]);
    } else if (j.stage == 1) {
        drop_redundant_v_avx2(j.K, c0, c1, j.Hc, 5, 1);
    } else {
        static thread_local std::vector<int16_t> blk;
        drop_redundant_u_block_avx2(j.K, j.Wc, j.Hc, 5, 1, 16 * part, blk);
    }
}

bool support_filter_team_usable(const sv_params &p) {
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    return have_avx2 && p.incon_window_size >= 0 && p.incon_window_size <= 7 && p.incon_threshold >= 0 && p.incon_threshold < 16384 && p.incon_min_support >= 1;
}

// The three filters of support_filter_t by a team; the filtered lattice ends up in T again.
static void filter_lattice_team(const sv_params &p, int16_t *T, int Wc, int Hc, const FilterTeam &team, FilterScratch &sc) {
    const size_t lat = (size_t)Wc * Hc;
    if (sc.kept.size() < lat + LATTICE_PAD) sc.kept.assign(lat + LATTICE_PAD, -1);
    const int parts = std::max(1, std::min(Wc, 16));
    if ((int)sc.undecided.size() < parts) sc.undecided.resize(parts);
    TeamJob job{&p, T, sc.kept.data(), Wc, Hc, parts, sc.undecided.data(), 0};
    team.run(team.ctx, parts, team_piece, &job);
    for (int q = 0; q < parts; q++) resolve_inconsistent_avx2(p, T, job.K, Wc, Hc, job.und[q].data(), job.und[q].size());
    job.stage = 1;
    team.run(team.ctx, parts, team_piece, &job);
    job.stage = 2;
    team.run(team.ctx, (Hc + 15) / 16, team_piece, &job);
    memcpy(T, job.K, sizeof(int16_t) * lat);
}
#else
bool support_filter_team_usable(const sv_params &) { return false; }
#endif

int support_filter_t(const sv_params &p, int16_t *T, int W, int H, int32_t *out, int cap, const FilterTeam *team, FilterScratch *scratch) {
    int Wc, Hc;
    lattice_dims(p, W, H, Wc, Hc);
#if defined(__x86_64__)
    if (team && scratch && team->threads > 1 && support_filter_team_usable(p)) {
        filter_lattice_team(p, T, Wc, Hc, *team, *scratch);
    } else
#endif
    {
        drop_inconsistent(p, T, Wc, Hc);
        drop_redundant(T, Wc, Hc, 5, 1, true);
        drop_redundant(T, Wc, Hc, 5, 1, false);
    }
    const int step = lattice_step(p);
    int n = 0;
#ifdef SV_FILTER_PROFILE
    const double tp0 = sv_prof_now();
#endif
    n = collect_points(T, Wc, Hc, step, out, cap);  // elas.cpp:424-428: u outer, v inner
#ifdef SV_FILTER_PROFILE
    const double tp1 = sv_prof_now();
    sv_prof_collect += tp1 - tp0;
    struct ProfEnd { double t0; ~ProfEnd() { sv_prof_corners += sv_prof_now() - t0; } } prof_end{tp1};
#endif
    if (p.add_corners) {  // elas.cpp:235-264
        if (n + 6 > cap) return -(n + 6);
        const int bu[4] = {0, 0, W - 1, W - 1}, bv[4] = {0, H - 1, 0, H - 1};
        int bd[4] = {0, 0, 0, 0};
        int best[4] = {10000000, 10000000, 10000000, 10000000};
        for (int j = 0; j < n; j++) {  // one pass for the four corners; strict '<' keeps the first of equally near points
            const int pu = out[3 * j], pv = out[3 * j + 1], pd = out[3 * j + 2];
            for (int i = 0; i < 4; i++) {
                const int du = bu[i] - pu, dv = bv[i] - pv;
                const int dist = du * du + dv * dv;
                const bool nearer = dist < best[i];
                best[i] = nearer ? dist : best[i];
                bd[i] = nearer ? pd : bd[i];
            }
        }
        const int base = n;
        for (int i = 0; i < 4; i++) {
            out[3 * n] = bu[i];
            out[3 * n + 1] = bv[i];
            out[3 * n + 2] = bd[i];
            n++;
        }
        for (int i = 2; i < 4; i++) {  // the two right-image corners (:258-259)
            out[3 * n] = out[3 * (base + i)] + out[3 * (base + i) + 2];
            out[3 * n + 1] = out[3 * (base + i) + 1];
            out[3 * n + 2] = out[3 * (base + i) + 2];
            n++;
        }
    } else if (n > cap) {
        return -n;
    }
    return n;
}

// row-major [Hc][Wc] front-end (the oracle's / reference's layout)
int support_filter(const sv_params &p, int16_t *dcan, int W, int H, int32_t *out, int cap) {
    int Wc, Hc;
    lattice_dims(p, W, H, Wc, Hc);
    std::vector<int16_t> T((size_t)Wc * Hc + LATTICE_PAD, 0);
    for (int vc = 0; vc < Hc; vc++)
        for (int uc = 0; uc < Wc; uc++) T[(size_t)uc * Hc + vc] = dcan[(size_t)vc * Wc + uc];
    const int n = support_filter_t(p, T.data(), W, H, out, cap);
    for (int vc = 0; vc < Hc; vc++)
        for (int uc = 0; uc < Wc; uc++) dcan[(size_t)vc * Wc + uc] = T[(size_t)uc * Hc + vc];
    return n;
}

// test hook: the team made of std::threads started per call (slow to start, same arithmetic)
int support_filter_threads(const sv_params &p, int16_t *dcan, int W, int H, int32_t *out, int cap, int threads) {
    int Wc, Hc;
    lattice_dims(p, W, H, Wc, Hc);
    std::vector<int16_t> T((size_t)Wc * Hc + LATTICE_PAD, 0);
    for (int vc = 0; vc < Hc; vc++)
        for (int uc = 0; uc < Wc; uc++) T[(size_t)uc * Hc + vc] = dcan[(size_t)vc * Wc + uc];
    struct Ctx {
        int threads;
    } ctx{std::max(1, threads)};
    FilterTeam team;
    team.ctx = &ctx;
    team.threads = ctx.threads;
    team.run = [](void *c, int parts, void (*fn)(void *, int), void *arg) {
        const int nt = static_cast<Ctx *>(c)->threads;
        std::atomic<int> next{0};
        auto work = [&] {
            for (int q = next.fetch_add(1); q < parts; q = next.fetch_add(1)) fn(arg, q);
        };
        std::vector<std::thread> th;
        for (int i = 1; i < nt; i++) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
    };
    FilterScratch sc;
    const int n = support_filter_t(p, T.data(), W, H, out, cap, &team, &sc);
    for (int vc = 0; vc < Hc; vc++)
        for (int uc = 0; uc < Wc; uc++) dcan[(size_t)vc * Wc + uc] = T[(size_t)uc * Hc + vc];
    return n;
}

// ------------------------------------------------------------------------------------------------------------
// Delaunay
// ------------------------------------------------------------------------------------------------------------

namespace {
// orientation arithmetic without table loads: next = {1, 2, 0}[o], prev = {2, 0, 1}[o]
inline int next3(int o) { return (9 >> (2 * o)) & 3; }
inline int prev3(int o) { return (18 >> (2 * o)) & 3; }
inline int32_t hnext(int32_t h) { return (h & ~3) | next3(h & 3); }
inline int32_t hprev(int32_t h) { return (h & ~3) | prev3(h & 3); }
}  // namespace

#define SYM(h) (T[(h) >> 2].nbr[(h)&3])
#define ORG(h) (T[(h) >> 2].vtx[next3((h)&3)])
#define DEST(h) (T[(h) >> 2].vtx[prev3((h)&3)])
#define APEX(h) (T[(h) >> 2].vtx[(h)&3])
#define BOND(a, b)                       \
    do {                                 \
        const int32_t a_ = (a), b_ = (b); \
        T[a_ >> 2].nbr[a_ & 3] = b_;     \
        T[b_ >> 2].nbr[b_ & 3] = a_;     \
    } while (0)
#define PX(v) ((int64_t)xy_[2 * (v)])
#define PY(v) ((int64_t)xy_[2 * (v) + 1])

static inline int64_t orient(const int32_t *xy, int a, int b, int c) {
    return ((int64_t)xy[2 * a] - xy[2 * c]) * ((int64_t)xy[2 * b + 1] - xy[2 * c + 1]) - ((int64_t)xy[2 * a + 1] - xy[2 * c + 1]) * ((int64_t)xy[2 * b] - xy[2 * c]);
}

static inline int64_t incirc(const int32_t *xy, int a, int b, int c, int d) {
    const int64_t adx = (int64_t)xy[2 * a] - xy[2 * d], ady = (int64_t)xy[2 * a + 1] - xy[2 * d + 1];
    const int64_t bdx = (int64_t)xy[2 * b] - xy[2 * d], bdy = (int64_t)xy[2 * b + 1] - xy[2 * d + 1];
    const int64_t cdx = (int64_t)xy[2 * c] - xy[2 * d], cdy = (int64_t)xy[2 * c + 1] - xy[2 * d + 1];
    return (adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) + (bdx * bdx + bdy * bdy) * (cdx * ady - adx * cdy) + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady);
}

uint32_t Delaunay::rnd(uint32_t choices) {  // triangle.cpp:3833-3836; seed < 714025 so 32-bit arithmetic is exact
    seed_ = (seed_ * 1366u + 150889u) % 714025u;
    return seed_ / (714025u / choices + 1);
}

Delaunay::H Delaunay::make(int &cursor) {  // triangle.cpp:2068-2101; slot 0 stands for "outer space"
    Tri &t = tris_[cursor];
    t.nbr[0] = t.nbr[1] = t.nbr[2] = 0;
    t.vtx[0] = t.vtx[1] = t.vtx[2] = -1;
    return (H)(cursor++) << 2;
}

// sort keys: x and y are biased to unsigned 16-bit and packed as (x << 16 | y); the (y, x) order is the key rotated by 16
constexpr int KEY_BIAS_X = 4096, KEY_BIAS_Y = 4096;
static inline uint32_t key_pack(int32_t x, int32_t y) { return ((uint32_t)(x + KEY_BIAS_X) << 16) | (uint32_t)(y + KEY_BIAS_Y); }
static inline uint32_t key_yx(uint32_t k) { return (k << 16) | (k >> 16); }

// Stable LSD radix sort of the packed keys, 11 bits per pass, through the `sorted_` scratch array.
void Delaunay::radix_sort_xy(Pt *a, int n) {
    if ((int)sorted_.size() < n) sorted_.resize(n);
    Pt *b = sorted_.data();
    uint32_t cnt[3][2048];
    memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < n; i++) {
        const uint32_t k = a[i].key;
        cnt[0][k & 2047]++;
        cnt[1][(k >> 11) & 2047]++;
        cnt[2][k >> 22]++;
    }
    for (int p = 0; p < 3; p++) {
        uint32_t run = 0;
        for (int d = 0; d < (p == 2 ? 1024 : 2048); d++) {
            const uint32_t c = cnt[p][d];
            cnt[p][d] = run;
            run += c;
        }
    }
    for (int i = 0; i < n; i++) b[cnt[0][a[i].key & 2047]++] = a[i];
    for (int i = 0; i < n; i++) a[cnt[1][(b[i].key >> 11) & 2047]++] = b[i];
    for (int i = 0; i < n; i++) b[cnt[2][a[i].key >> 22]++] = a[i];
    memcpy(a, b, sizeof(Pt) * (size_t)n);
}

// Randomised quicksort by (x, y) with the reference's pivot sequence (triangle.cpp:5183-5229): which of two coincident
// points survives the duplicate scan depends on it.
void Delaunay::sort_xy(Pt *a, int n) {
    if (n == 2) {
        if (a[0].key > a[1].key) std::swap(a[0], a[1]);
        return;
    }
    const int pivot = (int)rnd((uint32_t)n);
    const uint32_t pk = a[pivot].key;
    int left = -1, right = n;
    while (left < right) {
        do {
            left++;
        } while (left <= right && a[left].key < pk);
        do {
            right--;
        } while (left <= right && a[right].key > pk);
        if (left < right) std::swap(a[left], a[right]);
    }
    if (left > 1) sort_xy(a, left);
    if (right < n - 2) sort_xy(a + right + 1, n - right - 1);
}

// k-d ordering of the x-sorted vertex array.
//
// The reference re-orders the sorted array with randomised quickselects around medians of alternating axes
// (alternateaxes / vertexmedian, triangle.cpp:5243-5325).  Its result does not depend on the pivots: after the duplicate
// scan the (x, y) and (y, x) keys are both strict total orders, so every median cut is a unique set split, and the
// recursion only stops at groups of <= 3 vertices which it leaves sorted by (x, y).  The same arrangement is produced
// here without any selection: every node keeps its vertices in both orders (xs by (x, y), ys by (y, x)) and a cut is an
// index split of one list plus a stable, branch-free partition of the other.  Elements are (y-rank << 32 | x-rank).
void Delaunay::kd_order(uint64_t *xs, uint64_t *xalt, uint64_t *ys, uint64_t *yalt, int n, int axis, Pt *out) {
    if (n <= 3) {  // leaves (and the forced axis 0 of triangle.cpp:5313-5316): (x, y) order
        for (int i = 0; i < n; i++) out[i] = sorted_[(uint32_t)xs[i]];
        return;
    }
    const int divider = n >> 1;
    int li = 0, ri = divider;
    if (axis == 0) {
        const uint32_t pivot = (uint32_t)xs[divider];
        for (int i = 0; i < n; i++) {
            const uint64_t e = ys[i];
            const int l = (uint32_t)e < pivot;
            yalt[ri + ((li - ri) & -l)] = e;
            li += l;
            ri += 1 - l;
        }
        kd_order(xs, xalt, yalt, ys, divider, 1, out);
        kd_order(xs + divider, xalt + divider, yalt + divider, ys + divider, n - divider, 1, out + divider);
    } else {
        const uint32_t pivot = (uint32_t)(ys[divider] >> 32);
        for (int i = 0; i < n; i++) {
            const uint64_t e = xs[i];
            const int l = (uint32_t)(e >> 32) < pivot;
            xalt[ri + ((li - ri) & -l)] = e;
            li += l;
            ri += 1 - l;
        }
        kd_order(xalt, xs, ys, yalt, divider, 0, out);
        kd_order(xalt + divider, xs + divider, ys + divider, yalt + divider, n - divider, 0, out + divider);
    }
}

// a[0..m) is sorted by (x, y) without duplicates; rearranges it into the order the reference's alternating cuts leave
void Delaunay::alternate_cuts(Pt *a, int m) {
    if (m <= 3) return;  // triangle.cpp:5904-5913 leaves such an array as sorted
    if ((int)sorted_.size() < m) sorted_.resize(m);
    if ((int)kd_.size() < 4 * m) kd_.resize((size_t)4 * m);
    memcpy(sorted_.data(), a, sizeof(Pt) * (size_t)m);
    uint64_t *xs = kd_.data(), *xalt = xs + m, *ys = xalt + m, *yalt = ys + m;
    // (y, x) order = stable sort of the (x, y)-sorted array by y alone: two counting passes over the 16-bit biased y
    uint32_t cnt0[257] = {0}, cnt1[257] = {0};
    for (int i = 0; i < m; i++) {
        const uint32_t y = a[i].key & 0xFFFFu;
        cnt0[(y & 0xFF) + 1]++;
        cnt1[(y >> 8) + 1]++;
    }
    for (int i = 0; i < 256; i++) {
        cnt0[i + 1] += cnt0[i];
        cnt1[i + 1] += cnt1[i];
    }
    for (int i = 0; i < m; i++) xalt[cnt0[a[i].key & 0xFF]++] = ((uint64_t)(a[i].key & 0xFFFFu) << 32) | (uint32_t)i;
    for (int i = 0; i < m; i++) {
        const uint64_t e = xalt[i];
        yalt[cnt1[(e >> 40) & 0xFF]++] = e;
    }
    for (int i = 0; i < m; i++) {  // y-rank of every vertex
        const uint32_t p = (uint32_t)yalt[i];
        xs[p] = ((uint64_t)i << 32) | p;
        ys[i] = ((uint64_t)i << 32) | p;
    }
    kd_order(xs, xalt, ys, yalt, m, 0, a);
}

// triangle.cpp:5362-5651
void Delaunay::merge(H &farleft, H &innerleft, H &innerright, H &farright, int axis, int &cursor) {
    Tri *T = tris_.data();
    const int32_t *xy = xy_;
    int ild = DEST(innerleft), ila = APEX(innerleft);
    int iro = ORG(innerright), ira = APEX(innerright);
    if (axis == 1) {  // horizontal cut: walk the four extreme handles to the bottom-/top-most hull vertices
        int flp = ORG(farleft), fla = APEX(farleft);
        int frp = DEST(farright);
        while (PY(fla) < PY(flp)) {
            farleft = SYM(hnext(farleft));
            flp = fla;
            fla = APEX(farleft);
        }
        H chk = SYM(innerleft);
        int cv = APEX(chk);
        while (PY(cv) > PY(ild)) {
            innerleft = hnext(chk);
            ila = ild;
            ild = cv;
            chk = SYM(innerleft);
            cv = APEX(chk);
        }
        while (PY(ira) < PY(iro)) {
            innerright = SYM(hnext(innerright));
            iro = ira;
            ira = APEX(innerright);
        }
        chk = SYM(farright);
        cv = APEX(chk);
        while (PY(cv) > PY(frp)) {
            farright = hnext(chk);
            frp = cv;
            chk = SYM(farright);
            cv = APEX(chk);
        }
    }
    for (bool changed = true; changed;) {  // lower common tangent
        changed = false;
        if (orient(xy, ild, ila, iro) > 0) {
            innerleft = SYM(hprev(innerleft));
            ild = ila;
            ila = APEX(innerleft);
            changed = true;
        }
        if (orient(xy, ira, iro, ild) > 0) {
            innerright = SYM(hnext(innerright));
            iro = ira;
            ira = APEX(innerright);
            changed = true;
        }
    }
    H leftcand = SYM(innerleft), rightcand = SYM(innerright);
    H base = make(cursor);
    T = tris_.data();
    BOND(base, innerleft);
    base = hnext(base);
    BOND(base, innerright);
    base = hnext(base);
    ORG(base) = iro;
    DEST(base) = ild;
    if (ild == ORG(farleft)) farleft = hnext(base);
    if (iro == DEST(farright)) farright = hprev(base);
    int ll = ild, lr = iro;
    int ul = APEX(leftcand), ur = APEX(rightcand);
    for (;;) {
        const bool leftdone = orient(xy, ul, ll, lr) <= 0, rightdone = orient(xy, ur, ll, lr) <= 0;
        if (leftdone && rightdone) {
            H top = make(cursor);
            T = tris_.data();
            ORG(top) = ll;
            DEST(top) = lr;
            BOND(top, base);
            top = hnext(top);
            BOND(top, rightcand);
            top = hnext(top);
            BOND(top, leftcand);
            if (axis == 1) {  // back to left-/right-most handles
                int flp = ORG(farleft);
                int frp = DEST(farright), fra = APEX(farright);
                H chk = SYM(farleft);
                int cv = APEX(chk);
                while (PX(cv) < PX(flp)) {
                    farleft = hprev(chk);
                    flp = cv;
                    chk = SYM(farleft);
                    cv = APEX(chk);
                }
                while (PX(fra) > PX(frp)) {
                    farright = SYM(hprev(farright));
                    frp = fra;
                    fra = APEX(farright);
                }
            }
            return;
        }
        if (!leftdone) {  // flip away left edges that the circle through ll, lr, ul invalidates
            H nx = SYM(hprev(leftcand));
            int na = APEX(nx);
            if (na != -1) {
                bool bad = incirc(xy, ll, lr, ul, na) > 0;
                while (bad) {
                    nx = hnext(nx);
                    const H topc = SYM(nx);
                    nx = hnext(nx);
                    const H sidec = SYM(nx);
                    BOND(nx, topc);
                    BOND(leftcand, sidec);
                    leftcand = hnext(leftcand);
                    const H outerc = SYM(leftcand);
                    nx = hprev(nx);
                    BOND(nx, outerc);
                    ORG(leftcand) = ll;
                    DEST(leftcand) = -1;
                    APEX(leftcand) = na;
                    ORG(nx) = -1;
                    DEST(nx) = ul;
                    APEX(nx) = na;
                    ul = na;
                    nx = sidec;
                    na = APEX(nx);
                    bad = na != -1 && incirc(xy, ll, lr, ul, na) > 0;
                }
            }
        }
        if (!rightdone) {
            H nx = SYM(hnext(rightcand));
            int na = APEX(nx);
            if (na != -1) {
                bool bad = incirc(xy, ll, lr, ur, na) > 0;
                while (bad) {
                    nx = hprev(nx);
                    const H topc = SYM(nx);
                    nx = hprev(nx);
                    const H sidec = SYM(nx);
                    BOND(nx, topc);
                    BOND(rightcand, sidec);
                    rightcand = hprev(rightcand);
                    const H outerc = SYM(rightcand);
                    nx = hnext(nx);
                    BOND(nx, outerc);
                    ORG(rightcand) = -1;
                    DEST(rightcand) = lr;
                    APEX(rightcand) = na;
                    ORG(nx) = ur;
                    DEST(nx) = -1;
                    APEX(nx) = na;
                    ur = na;
                    nx = sidec;
                    na = APEX(nx);
                    bad = na != -1 && incirc(xy, ll, lr, ur, na) > 0;
                }
            }
        }
        if (leftdone || (!rightdone && incirc(xy, ul, ll, lr, ur) > 0)) {
            BOND(base, rightcand);
            base = hprev(rightcand);
            DEST(base) = ll;
            lr = ur;
            rightcand = SYM(base);
            ur = APEX(rightcand);
        } else {
            BOND(base, leftcand);
            base = hnext(leftcand);
            ORG(base) = lr;
            ll = ul;
            leftcand = SYM(base);
            ul = APEX(leftcand);
        }
    }
}

// triangle.cpp:5670-5815
void Delaunay::build(const Pt *p, int n, int axis, H &farleft, H &farright, int &cursor) {
    const int a[3] = {p[0].id, p[1].id, n > 2 ? p[2].id : -1};
    if (n == 2) {
        H l = make(cursor), r = make(cursor);
        Tri *T = tris_.data();
        ORG(l) = a[0];
        DEST(l) = a[1];
        ORG(r) = a[1];
        DEST(r) = a[0];
        BOND(l, r);
        l = hprev(l);
        r = hnext(r);
        BOND(l, r);
        l = hprev(l);
        r = hnext(r);
        BOND(l, r);
        farright = r;
        farleft = hprev(r);
    } else if (n == 3) {
        H mid = make(cursor), t1 = make(cursor), t2 = make(cursor), t3 = make(cursor);
        Tri *T = tris_.data();
        const int64_t area = orient(xy_, a[0], a[1], a[2]);
        if (area == 0) {
            ORG(mid) = a[0];
            DEST(mid) = a[1];
            ORG(t1) = a[1];
            DEST(t1) = a[0];
            ORG(t2) = a[2];
            DEST(t2) = a[1];
            ORG(t3) = a[1];
            DEST(t3) = a[2];
            BOND(mid, t1);
            BOND(t2, t3);
            mid = hnext(mid);
            t1 = hprev(t1);
            t2 = hnext(t2);
            t3 = hprev(t3);
            BOND(mid, t3);
            BOND(t1, t2);
            mid = hnext(mid);
            t1 = hprev(t1);
            t2 = hnext(t2);
            t3 = hprev(t3);
            BOND(mid, t1);
            BOND(t2, t3);
            farleft = t1;
            farright = t2;
        } else {
            const int second = area > 0 ? a[1] : a[2], third = area > 0 ? a[2] : a[1];
            ORG(mid) = a[0];
            DEST(t1) = a[0];
            ORG(t3) = a[0];
            DEST(mid) = second;
            ORG(t1) = second;
            DEST(t2) = second;
            APEX(mid) = third;
            ORG(t2) = third;
            DEST(t3) = third;
            BOND(mid, t1);
            mid = hnext(mid);
            BOND(mid, t2);
            mid = hnext(mid);
            BOND(mid, t3);
            t1 = hprev(t1);
            t2 = hnext(t2);
            BOND(t1, t2);
            t1 = hprev(t1);
            t3 = hprev(t3);
            BOND(t1, t3);
            t2 = hnext(t2);
            t3 = hprev(t3);
            BOND(t2, t3);
            farleft = t1;
            farright = area > 0 ? t2 : hnext(farleft);
        }
    } else {
        const int divider = n >> 1;
        H innerleft, innerright;
        build(p, divider, 1 - axis, farleft, innerleft, cursor);
        build(p + divider, n - divider, 1 - axis, innerright, farright, cursor);
        merge(farleft, innerleft, innerright, farright, axis, cursor);
    }
}

// Slots a subproblem of n vertices allocates: 2 for a pair, 4 for a triple, 2 per merge.  It only depends on n, so the two
// halves of the top-level cut can be built at the same time into disjoint, pre-computed slot ranges and still leave the
// pool exactly as the sequential recursion does (the output order is the pool order).
static int slots_of(int n) { return n == 2 ? 2 : n == 3 ? 4 : slots_of(n >> 1) + slots_of(n - (n >> 1)) + 2; }

// Sort, duplicate scan and k-d ordering: fills order_[0..m) and returns m (< 2: nothing to triangulate; -2: coordinates
// outside the packed-key range).
int Delaunay::prepare(const int32_t *xy, int n) {
    xy_ = xy;
    seed_ = 1;  // triangle.cpp:3818: reseeded on every call
    if ((int)order_.size() < n) order_.resize(n);
    Pt *a = order_.data();
    for (int i = 0; i < n; i++) {
        const int32_t x = xy[2 * i], y = xy[2 * i + 1];
        if (x < -KEY_BIAS_X || x >= 65536 - KEY_BIAS_X || y < -KEY_BIAS_Y || y >= 65536 - KEY_BIAS_Y) return -2;  // outside the packed-key range
        a[i] = Pt{key_pack(x, y), i};
    }
    // Sorted (x, y) order.  Without coincident points it is unique, whatever the sorting method: a stable LSD radix sort (three
    // 11-bit digits of the packed key) is 3x faster than the reference's randomised quicksort.  Only when two points coincide
    // does the quicksort's pivot sequence decide which of them comes first and survives the duplicate scan below
    // (triangle.cpp:5183-5229, :5890-5903) - then the array is rebuilt and sorted the reference's way.  Coincident points only
    // arise in the right image (u - d equal on one row) and are rare.
    radix_sort_xy(a, n);
    bool coincident = false;
    for (int j = 1; j < n; j++) coincident |= a[j].key == a[j - 1].key;
    if (coincident) {
        for (int i = 0; i < n; i++) a[i] = Pt{key_pack(xy[2 * i], xy[2 * i + 1]), i};
        sort_xy(a, n);
    }
    int m = 0;
    for (int j = 1; j < n; j++) {  // triangle.cpp:5890-5903: the first of a group of coincident points is kept
        if (a[m].key == a[j].key) continue;
        a[++m] = a[j];
    }
    m++;
    if (m >= 2) alternate_cuts(a, m);
    return m;
}

// The vertex ids in the order the divide-and-conquer recursion consumes them (for the GPU triangulation, delaunay_gpu.hip).
int Delaunay::kd_ordered_ids(const int32_t *xy, int n, int32_t *ids_out) {
    if (n < 3) return 0;
    const int m = prepare(xy, n);
    for (int i = 0; i < m; i++) ids_out[i] = order_[i].id;
    return m;
}

// build() with the right half of the cut handed to another thread, `depth` levels deep (depth 2: four quarters on four threads).
// The slots a subproblem allocates depend on its size only (slots_of), so the halves fill disjoint, pre-computed slot ranges
// and the pool ends up exactly as the sequential recursion leaves it.  A half is claimed with a compare-and-swap: it runs
// exactly once, by the helper or - if nobody picked it up in time - by the spawning thread itself, so nobody can wait for ever.
// The hand-over record is shared by the two parties and freed by whoever lets go of it last: a helper that shows up after
// the spawner has moved on still finds valid memory (and nothing left to do).
void Delaunay::build_split(const Pt *p, int n, int axis, H &farleft, H &farright, int cursor0, int depth, const Spawn *spawn, int &cursor_end) {
    if (depth <= 0 || n < 64) {
        int c = cursor0;
        build(p, n, axis, farleft, farright, c);
        cursor_end = c;
        return;
    }
    const int divider = n >> 1;
    struct Half {
        Delaunay *self;
        const Pt *p;
        const Spawn *spawn;
        int n, axis, cursor0, depth, cursor_end;
        H l, r;
        std::atomic<int> state{0};  // 0 queued, 1 claimed, 2 done
        std::atomic<int> refs{2};
        void work() {
            int expected = 0;
            if (!state.compare_exchange_strong(expected, 1)) return;
            self->build_split(p, n, axis, l, r, cursor0, depth, spawn, cursor_end);
            state.store(2, std::memory_order_release);
        }
        void release() {
            if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this;
        }
        static void run(void *arg) {  // the helper's entry point
            Half *h = static_cast<Half *>(arg);
            h->work();
            h->release();
        }
    };
    Half *right = new Half();
    right->self = this;
    right->p = p + divider;
    right->spawn = spawn;  // (outlives the call: a late helper finds state != 0 and never touches it)
    right->n = n - divider;
    right->axis = 1 - axis;
    right->cursor0 = cursor0 + slots_of(divider);
    right->depth = depth - 1;
    right->cursor_end = 0;
    right->l = right->r = 0;
    spawn->run(spawn->ctx, &Half::run, right);
    H innerleft;
    int left_end = cursor0;
    build_split(p, divider, 1 - axis, farleft, innerleft, cursor0, depth - 1, spawn, left_end);
    right->work();  // no-op unless it is still unclaimed
    while (right->state.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
    int c = right->cursor_end;
    H rl = right->l, rr = right->r;
    right->release();
    merge(farleft, innerleft, rl, rr, axis, c);
    farright = rr;
    cursor_end = c;
}

int Delaunay::triangulate(const int32_t *xy, int n, int32_t *tri_out, int cap, const Spawn *spawn) {
    if (n < 3) return 0;
    // leaves allocate <= 4 slots per 3 points (2 per 2), every merge 2 more: < 3n in total, + the outer-space slot
    if ((int)tris_.size() < 3 * n + 8) tris_.resize(3 * n + 8);
    const int m = prepare(xy, n);
    if (m < 0) return m;
    if (m < 2) return 0;
    n_slots_ = 0;
    make(n_slots_);
    Pt *a = order_.data();
    H hl, hr;
    if (spawn && spawn->run && m >= 64) {
        // latency mode: the recursion's top levels are shared with other threads (build_split)
        int end = n_slots_;
        build_split(a, m, 0, hl, hr, n_slots_, spawn->depth > 0 ? spawn->depth : 1, spawn, end);
        n_slots_ = end;
    } else {
        build(a, m, 0, hl, hr, n_slots_);
    }
    // Output in slot order (= pool order, triangle.cpp:7449-7500) skipping bounding triangles (what removeghosts,
    // :5817-5859, deletes): corners are org/dest/apex at orientation 0.
    int count = 0;
    const Tri *T = tris_.data();
    for (int t = 1; t < n_slots_; t++) {
        if (T[t].vtx[0] < 0 || T[t].vtx[1] < 0 || T[t].vtx[2] < 0) continue;
        if (count >= cap) return -1;
        tri_out[3 * count] = T[t].vtx[1];
        tri_out[3 * count + 1] = T[t].vtx[2];
        tri_out[3 * count + 2] = T[t].vtx[0];
        count++;
    }
    return count;
}

}  // namespace sv
