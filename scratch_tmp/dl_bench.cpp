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
    // points: file of int32 triples (u,v,d)
    FILE *f = fopen(argv[1], "rb");
    std::vector<int32_t> sup;
    int32_t b[3];
    while (fread(b, 4, 3, f) == 3) { sup.push_back(b[0]); sup.push_back(b[1]); sup.push_back(b[2]); }
    fclose(f);
    int n = sup.size() / 3;
    std::vector<int32_t> xy(2 * n), tri(6 * n * 3);
    Delaunay dl;
    double tt[2] = {0, 0};
    int iters = 2000;
    long cnt = 0;
    for (int side = 0; side < 2; side++) {
        for (int q = 0; q < n; q++) { xy[2 * q] = side ? sup[3 * q] - sup[3 * q + 2] : sup[3 * q]; xy[2 * q + 1] = sup[3 * q + 1]; }
        double t0 = now();
        for (int it = 0; it < iters; it++) cnt += dl.triangulate(xy.data(), n, tri.data(), 2 * n);
        tt[side] = (now() - t0) / iters;
    }
    printf("n=%d  left %.1f us  right %.1f us  (tris %ld)\n", n, tt[0] * 1e6, tt[1] * 1e6, cnt / iters);
    return 0;
}
