// How long a chain of N small dependent kernels takes on one stream, launched one by one against launched as one hipGraph
// (single-pair latency: phase 2 is 13 dependent launches of 5 - 16 us).   hipcc -O2 --offload-arch=gfx950 tools/graph_chain.hip -o abl_tmp/graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_step(float *p, int n, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = i < n ? p[i] : 0.f;
    for (int s = 0; s < spin; s++) v = v * 1.0001f + 0.5f;
    if (i < n) p[i] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 13, blocks = argc > 2 ? atoi(argv[2]) : 512, spin = argc > 3 ? atoi(argv[3]) : 200, reps = 300;
    float *d;
    const int n = blocks * 256;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<double> a, b, c;
    for (int r = 0; r < reps; r++) {  // one by one
        const double t0 = now_us();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, d, n, spin);
        CK(hipStreamSynchronize(st));
        a.push_back(now_us() - t0);
    }
    {  // one kernel alone, for the floor
        for (int r = 0; r < reps; r++) {
            const double t0 = now_us();
            hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, d, n, spin);
            CK(hipStreamSynchronize(st));
            c.push_back(now_us() - t0);
        }
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_step, dim3(blocks), dim3(256), 0, st, d, n, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < reps; r++) {
        const double t0 = now_us();
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        b.push_back(now_us() - t0);
    }
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%d dependent kernels of %d workgroups (spin %d): one by one %.1f us, as a graph %.1f us; one kernel alone %.1f us\n", N, blocks, spin, med(a), med(b), med(c));
    return 0;
}
