// Which engine moves a device -> page-locked-host copy while kernels and uploads run?  (ROCm 7.2: the runtime falls back to a shader
// copy, `__amd_rocclr_copyBuffer`, when it finds the DMA engines busy - seen as 13 % of the kernel time of the host-memory path.)
//   hipcc --offload-arch=gfx950 -O2 tools/copy_path.hip -o abl_tmp/copy_path
//   abl_tmp/copy_path <variant 0..3> <busy 0/1> <upload 0/1>     variant: 0 hipMemcpyAsync, 1 hipMemcpyBatchAsync + PreferOverlapWithCompute,
//                                                                 (not supported in 7.2), 2 hipMemcpyDtoHAsync, 3 own copy kernel writing host memory,
//                                                                 4 hipMemcpyAsync behind a stream wait on a kernel of another stream, 5 behind a host wait
// prints the mean time of a 30 MB download; run under `rocprofv3 --kernel-trace --stats` to see whether copy kernels appear.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(float *p, int iters) {
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) p[0] = a;
}
__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0, busy = argc > 2 ? atoi(argv[2]) : 1, upload = argc > 3 ? atoi(argv[3]) : 1;
    const size_t MB = 1 << 20, dn = 30 * MB, un = 60 * MB;
    char *d_src, *d_up, *h_dst, *h_up;
    float *d_busy;
    CK(hipMalloc(&d_src, dn));
    CK(hipMalloc(&d_up, un));
    CK(hipMalloc(&d_busy, 4096));
    CK(hipHostMalloc(&h_dst, dn));
    CK(hipHostMalloc(&h_up, un));
    CK(hipMemset(d_src, 1, dn));
    hipStream_t sk[4], sc, su;
    for (auto &s : sk) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double total = 0;
    const int reps = 20;
    for (int r = 0; r < reps + 2; r++) {
        if (busy)
            for (auto &s : sk) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s, d_busy, 60000);  // ~ms of full-chip VALU work per stream
        if (upload) CK(hipMemcpyAsync(d_up, h_up, un, hipMemcpyHostToDevice, su));
        if (variant == 4 || variant == 5) {  // the download depends on a kernel of another stream: by a stream wait (4) or a host wait (5)
            hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, sk[0], d_busy, 1000);
            CK(hipEventRecord(e2, sk[0]));
            if (variant == 4) CK(hipStreamWaitEvent(sc, e2, 0));
            else CK(hipEventSynchronize(e2));
        }
        CK(hipEventRecord(e0, sc));
        if (variant == 0 || variant == 4 || variant == 5) {
            CK(hipMemcpyAsync(h_dst, d_src, dn, hipMemcpyDeviceToHost, sc));
        } else if (variant == 1) {
            void *dsts[1] = {h_dst}, *srcs[1] = {d_src};
            size_t sizes[1] = {dn}, idx[1] = {0}, fail = 0;
            hipMemcpyAttributes at = {};
            at.srcAccessOrder = hipMemcpySrcAccessOrderStream;
            at.srcLocHint.type = hipMemLocationTypeDevice, at.srcLocHint.id = 0;
            at.dstLocHint.type = hipMemLocationTypeHost, at.dstLocHint.id = 0;
            at.flags = hipMemcpyFlagPreferOverlapWithCompute;
            CK(hipMemcpyBatchAsync(dsts, srcs, sizes, 1, &at, idx, 1, &fail, sc));
        } else if (variant == 2) {
            CK(hipMemcpyDtoHAsync(h_dst, (hipDeviceptr_t)d_src, dn, sc));
        } else {
            hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, sc, (const uint4 *)d_src, (uint4 *)h_dst, dn / 16);
        }
        CK(hipEventRecord(e1, sc));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) total += ms;
        CK(hipDeviceSynchronize());
    }
    printf("variant %d busy %d upload %d: 30 MB download %.3f ms (%.1f GB/s), first bytes %d\n", variant, busy, upload, total / reps, 30.0 * MB / (total / reps) / 1e6, (int)h_dst[0] + (int)h_dst[dn - 1]);
    return 0;
}
