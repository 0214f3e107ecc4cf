// PCIe rates of OUR OWN copy kernels between page-locked host memory and HBM (the alternative to the DMA engines for the host-memory
// path): workgroups needed, both directions at once, stream priority, and what a chip full of SHORT compute workgroups costs them.
//   hipcc --offload-arch=gfx950 -O2 tools/zc_rate.hip -o abl_tmp/zc_rate && abl_tmp/zc_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(float *p, int iters) {  // a short workgroup (tens of us), launched by the thousand
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) p[0] = a;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base < n; base += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (base + (size_t)u * 256 < n) v[u] = __builtin_nontemporal_load(src + base + (size_t)u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (base + (size_t)u * 256 < n) __builtin_nontemporal_store(v[u], dst + base + (size_t)u * 256);
    }
}

int main(int argc, char **argv) {
    const size_t MB = 1 << 20, up = 60 * MB, dn = 119 * MB;  // a 64-pair chunk: 2 x 29.8 MB of gray rows up, 119 MB of D1 down
    char *d_up, *d_dn, *h_up, *h_dn;
    float *d_busy;
    CK(hipMalloc(&d_up, up));
    CK(hipMalloc(&d_dn, dn));
    CK(hipMalloc(&d_busy, 4096));
    CK(hipHostMalloc(&h_up, up));
    CK(hipHostMalloc(&h_dn, dn));
    CK(hipMemset(d_dn, 1, dn));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t sk[4], su, sd, suh, sdh;
    for (auto &s : sk) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&suh, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&sdh, hipStreamNonBlocking, hi));
    hipEvent_t e0, e1, f0, f1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&f0));
    CK(hipEventCreate(&f1));
    const int reps = 10;
    // how: 0 DMA engines (hipMemcpyAsync), 1 copy kernels, 2 copy kernels on high-priority streams
    for (int busy = 0; busy < 2; busy++)
        for (int how = 0; how < 3; how++)
            for (int wgs : {8, 16, 32, 64, 128, 512}) {
                if (how == 0 && wgs != 8) continue;
                for (int dir = 1; dir <= 3; dir++) {  // 1 up, 2 down, 3 both
                    double tu = 0, td = 0;
                    hipStream_t cu = how == 2 ? suh : su, cd = how == 2 ? sdh : sd;
                    for (int r = 0; r < reps + 2; r++) {
                        if (busy)
                            for (auto &s : sk)
                                for (int q = 0; q < 24; q++) hipLaunchKernelGGL(k_busy, dim3(8192), dim3(256), 0, s, d_busy, 4000);
                        if (dir & 1) {
                            CK(hipEventRecord(e0, cu));
                            if (how == 0) CK(hipMemcpyAsync(d_up, h_up, up, hipMemcpyHostToDevice, cu));
                            else hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, cu, (const u32x4 *)h_up, (u32x4 *)d_up, up / 16);
                            CK(hipEventRecord(e1, cu));
                        }
                        if (dir & 2) {
                            CK(hipEventRecord(f0, cd));
                            if (how == 0) CK(hipMemcpyAsync(h_dn, d_dn, dn, hipMemcpyDeviceToHost, cd));
                            else hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, cd, (const u32x4 *)d_dn, (u32x4 *)h_dn, dn / 16);
                            CK(hipEventRecord(f1, cd));
                        }
                        float ms = 0;
                        if (dir & 1) {
                            CK(hipEventSynchronize(e1));
                            CK(hipEventElapsedTime(&ms, e0, e1));
                            if (r >= 2) tu += ms;
                        }
                        if (dir & 2) {
                            CK(hipEventSynchronize(f1));
                            CK(hipEventElapsedTime(&ms, f0, f1));
                            if (r >= 2) td += ms;
                        }
                        CK(hipDeviceSynchronize());
                    }
                    printf("busy %d how %d wgs %3d dir %d:", busy, how, wgs, dir);
                    if (dir & 1) printf("  up %.3f ms %.1f GB/s", tu / reps, up / (tu / reps) / 1e6);
                    if (dir & 2) printf("  down %.3f ms %.1f GB/s", td / reps, dn / (td / reps) / 1e6);
                    printf("\n");
                    fflush(stdout);
                }
            }
    printf("check %d\n", (int)h_dn[0] + (int)h_dn[dn - 1]);
    return 0;
}
