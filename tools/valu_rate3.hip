// Micro-benchmark, third set: v_cndmask_b32 and the compare -> select pairs around it (tools/valu_rate2.hip measured 23 cycles for a
// bare v_cndmask_b32 reading vcc; this separates the mask source, the operand kinds and the hazards).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -w tools/valu_rate3.hip -o /tmp/valu_rate3 && /tmp/valu_rate3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define REP 64
template <int OP>
__global__ __launch_bounds__(1024) void kern(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    uint32_t b = seed ^ threadIdx.x, c = seed + 7;
    uint64_t m = 0x5555AAAA3333CCCCull * (seed | 1);
    uint4 q[4];
    for (int i = 0; i < 4; i++) q[i] = make_uint4(a[i], a[i + 1], a[(i + 2) & 7], a[(i + 3) & 7]);
    __shared__ uint4 lds[1024];
    lds[threadIdx.x] = q[0];
    __syncthreads();
    uint32_t addr = (threadIdx.x & 1023) * 16;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b));
                if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "s"(m));
                if (OP == 2) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 3) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_cmp_lt_i32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 6) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == 7) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b));
                if (OP == 8) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20");
                if (OP == 9) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q[i & 3]) : "v"(addr));
                if (OP == 10) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(addr));
                if (OP == 11) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_max_u32 %0, 0, %0" : "+v"(a[i]));
                if (OP == 13) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 14) asm volatile("v_subrev_co_u32 %0, vcc, 1, %0" : "+v"(a[i]) : : "vcc");
                if (OP == 15) asm volatile("v_cmp_eq_u32 vcc, 0, %0\n s_or_b64 s[20:21], vcc, s[20:21]" : : "v"(a[i]) : "vcc", "s20", "s21");
                if (OP == 16) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 17) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
                if (OP == 18) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(a[i]));
                if (OP == 20) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 21) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
                if (OP == 22) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
                if (OP == 23) asm volatile("v_add_u32 %0, -1, %0" : "+v"(a[i]));
            }
        }
    }
    uint32_t s = (uint32_t)m;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += q[i].x + q[i].y + q[i].z + q[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int waves_per_simd) {
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    const int threads = 64 * 4 * waves_per_simd;
    const int iters = 5000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<OP><<<256, threads>>>(out, 100, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<OP><<<256, threads>>>(out, iters, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * REP * waves_per_simd;
    printf("%-34s waves/SIMD=%d  %.3f ms  -> %.2f cycles per (group of) instruction(s) at 2.4 GHz\n", name, waves_per_simd, ms, ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;  // one operation per process (tools/valu_rate3.sh runs them under a timeout each)
    static const char *names[24] = {"v_cndmask vcc (a,b)", "v_cndmask_e64 sgpr mask", "v_cmp + v_cndmask", "v_cndmask vcc (0,a)", "v_cndmask vcc dst!=src", "v_cmp + s_nop 1 + v_cndmask",
                                    "v_cmp vcc", "v_cmp_e64 sgpr", "v_readfirstlane", "ds_read_b128 + wait", "ds_read_b128 (no wait)", "v_max_i32", "v_max_u32 const", "v_addc_co_u32",
                                    "v_subrev_co_u32", "v_cmp + s_or_b64", "v_perm_b32", "v_cvt_i32_f32", "v_lshlrev_b32 (reg shift)", "v_lshlrev_b32 (const 4)", "v_lshrrev_b32 (reg shift)",
                                    "v_not_b32", "v_bfe_u32", "v_add_u32 const"};
    for (int w : {2, 4}) {
#define R(i) if (only < 0 || only == i) run<i>(names[i], w);
        R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23)
    }
    return 0;
}
