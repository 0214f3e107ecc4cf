// Micro-benchmark: issue cost of integer / fp32 VALU instructions on gfx950, in cycles per wave-instruction per SIMD.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -w tools/valu_rate.hip -o abl_tmp/valu_rate && ./abl_tmp/valu_rate
// Result on MI355X: v_sad_u8, v_min_u32, v_med3_u32, v_lshl_or_b32, v_add3_u32 cost 4 cycles, v_add_u32, v_and_b32, v_mov_b32
// and v_fma_f32 2 cycles once two wavefronts share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
template <int OP>
__global__ __launch_bounds__(1024) void kern(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    uint32_t b = seed ^ threadIdx.x, c = seed + 7;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)a[i];
    uint64_t q[5];
    for (int i = 0; i < 5; i++) q[i] = ((uint64_t)a[i] << 32) | a[i + 1];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 2) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 3) asm volatile("v_med3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7]));
                if (OP == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 6) asm volatile("v_max_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 7) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 8) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
                // sliding-window forms: 4 SADs of the 4 bytes of src1 against byte offsets 0..3 of the 8 bytes of src0
                if (OP == 10) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i & 3]) : "v"(q[4]), "v"(c));
                if (OP == 11) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i & 3]) : "v"(q[4]), "v"(c));
                if (OP == 12) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 14) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 15) asm volatile("v_alignbyte_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 16) asm volatile("v_bfe_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 17) asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 18) asm volatile("v_pk_min_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)f[i];
    for (int i = 0; i < 5; i++) s += (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int waves_per_simd) {
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<OP><<<256, threads>>>(out, 100, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<OP><<<256, threads>>>(out, iters, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * REP * waves_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_sad_u8", w); run<1>("v_min_u32", w); run<2>("v_lshl_or_b32", w); run<3>("v_med3_u32", w); run<4>("v_fma_f32", w);
        run<5>("v_add_u32", w); run<6>("v_max_i32", w); run<7>("v_and_b32", w); run<8>("v_add3_u32", w); run<9>("v_mov_b32", w);
        run<10>("v_qsad_pk_u16_u8", w); run<11>("v_mqsad_pk_u16_u8", w); run<12>("v_msad_u8", w); run<13>("v_sad_hi_u8", w); run<14>("v_perm_b32", w);
        run<15>("v_alignbyte_b32", w); run<16>("v_bfe_u32", w); run<17>("v_pk_add_u16", w); run<18>("v_pk_min_u16", w); run<19>("v_lshlrev_b32", w);
    }
    return 0;
}
