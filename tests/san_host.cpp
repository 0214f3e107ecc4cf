// Sanitizer harness for the host stage (tests/test_sanitizers.py): split triangulations with early and late helpers against
// the sequential result, and the vectorised lattice filters on random lattices (they read up to LATTICE_PAD elements past the end).
#include "host_stage.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
using namespace sv;
#include <mutex>
struct Helper {
    std::vector<std::thread> threads;
    std::mutex mu;  // at depth > 1 helpers spawn helpers
    int delay_us;
    static void run(void *ctx, void (*fn)(void *), void *arg) {
        Helper *hp = static_cast<Helper *>(ctx);
        const int delay = hp->delay_us;
        std::lock_guard<std::mutex> lk(hp->mu);
        hp->threads.emplace_back([fn, arg, delay] {
            if (delay > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay));
            fn(arg);
        });
    }
};
int main() {
    std::mt19937 rng(5);
    int bad = 0;
    for (int it = 0; it < 300; it++) {
        int n = 64 + rng() % 3000;
        std::vector<int32_t> xy(2 * n);
        for (int i = 0; i < n; i++) { xy[2 * i] = (int)(rng() % 250) * 5 - (int)(rng() % 3) * 40; xy[2 * i + 1] = (int)(rng() % 75) * 5; }
        std::vector<int32_t> a(6 * n + 24), b(6 * n + 24);
        Delaunay d1, d2;
        int na = d1.triangulate(xy.data(), n, a.data(), 2 * n + 8);
        Helper h{{}, {}, (it % 3 == 0) ? 3000 : 0};
        Delaunay::Spawn sp{&Helper::run, &h, 1 + it % 3};  // halves, quarters, eighths
        int nb = d2.triangulate(xy.data(), n, b.data(), 2 * n + 8, &sp);
        for (size_t i = 0;; i++) {  // joined by index: the vector may still grow while late helpers start
            std::thread t;
            {
                std::lock_guard<std::mutex> lk(h.mu);
                if (i >= h.threads.size()) break;
                t = std::move(h.threads[i]);
            }
            t.join();
        }
        if (na != nb || memcmp(a.data(), b.data(), sizeof(int32_t) * 3 * na)) bad++;
    }
    // lattice filter on random lattices (padding respected)
    for (int it = 0; it < 200; it++) {
        int W = 40 + rng() % 600, H = 40 + rng() % 300;
        sv_params p; memset(&p, 0, sizeof(p));
        p.candidate_stepsize = 3 + rng() % 6; p.incon_window_size = rng() % 8; p.incon_threshold = 1 + rng() % 7; p.incon_min_support = 1 + rng() % 11; p.add_corners = rng() % 2;
        int step = p.candidate_stepsize, Wc = (W + step - 1) / step, Hc = (H + step - 1) / step;
        std::vector<int16_t> T((size_t)Wc * Hc + LATTICE_PAD, 0);
        for (int i = 0; i < Wc * Hc; i++) T[i] = (rng() % 3) ? (int16_t)(rng() % 60) : (int16_t)-1;
        std::vector<int32_t> out(3 * ((size_t)Wc * Hc + 6));
        // the same lattice (row-major copy) through a team of threads: same list, same lattice - and no thread reads what another writes
        std::vector<int16_t> rm((size_t)Wc * Hc), rm2;
        for (int vc = 0; vc < Hc; vc++)
            for (int uc = 0; uc < Wc; uc++) rm[(size_t)vc * Wc + uc] = T[(size_t)uc * Hc + vc];
        rm2 = rm;
        std::vector<int32_t> out_a(out.size()), out_b(out.size());
        const int na = support_filter(p, rm.data(), W, H, out_a.data(), Wc * Hc + 6);
        const int nb = support_filter_threads(p, rm2.data(), W, H, out_b.data(), Wc * Hc + 6, 2 + it % 4);
        if (na != nb || (na > 0 && memcmp(out_a.data(), out_b.data(), sizeof(int32_t) * 3 * na)) || rm != rm2) bad++;
        support_filter_t(p, T.data(), W, H, out.data(), Wc * Hc + 6);
    }
    printf("sanitizer run done, delaunay / filter-team mismatches: %d\n", bad);
    return bad != 0;
}
