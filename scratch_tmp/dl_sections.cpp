#define private public
#include "../low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd/csrc/host_stage.h"
#undef private
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace sv;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    std::vector<int32_t> sup;
    int32_t b[3];
    while (fread(b, 4, 3, f) == 3) { sup.push_back(b[0]); sup.push_back(b[1]); sup.push_back(b[2]); }
    fclose(f);
    int n = sup.size() / 3;
    std::vector<int32_t> xy(2 * n);
    for (int q = 0; q < n; q++) { xy[2 * q] = sup[3 * q]; xy[2 * q + 1] = sup[3 * q + 1]; }
    Delaunay dl;
    dl.order_.resize(n); dl.tris_.resize(3 * n + 8);
    double t[4] = {0,0,0,0};
    int iters = 2000;
    for (int it = 0; it < iters; it++) {
        dl.xy_ = xy.data(); dl.seed_ = 1; dl.n_slots_ = 0; dl.make();
        Delaunay::Pt *a = dl.order_.data();
        double t0 = now();
        for (int i = 0; i < n; i++) a[i] = Delaunay::Pt{((uint32_t)(xy[2*i]+4096) << 16) | (uint32_t)(xy[2*i+1]+4096), i};
        dl.sort_xy(a, n);
        double t1 = now();
        int m = n;
        dl.alternate_cuts(a, m);
        double t2 = now();
        Delaunay::H hl, hr;
        dl.build(a, m, 0, hl, hr);
        double t3 = now();
        t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2;
    }
    printf("n=%d sort %.1f us  alternate %.1f us  build %.1f us  slots %d\n", n, t[0]/iters*1e6, t[1]/iters*1e6, t[2]/iters*1e6, dl.n_slots_);
    return 0;
}
