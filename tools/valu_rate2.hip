// Micro-benchmark, second set: issue cost (cycles per wave64 instruction per SIMD) of the float / packed-float / mask / multiply
// instructions the round-4 kernel work chooses between, and the latency of a dependent LDS access (the GPU triangulation's seam walk).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -w tools/valu_rate2.hip -o abl_tmp/valu_rate2 && ./abl_tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void kern(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + i + 1);
    uint32_t b = seed ^ threadIdx.x, c = seed + 7;
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = (float)a[i];
    float fb = (float)b, fc = 0.25f;
    uint64_t q[8];
    for (int i = 0; i < 8; i++) q[i] = ((uint64_t)a[i] << 32) | a[(i + 1) & 7];
    float2v p[8];
    for (int i = 0; i < 8; i++) p[i] = float2v{f[i], f[(i + 3) & 7]};
    float2v pb = float2v{fb, fc};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "v"(fb));
                if (OP == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "v"(fb));
                if (OP == 2) asm volatile("v_max_f32 %0, %1, %0" : "+v"(f[i]) : "v"(fb));
                if (OP == 3) asm volatile("v_fma_f32 %0, %1, %2, %0 clamp" : "+v"(f[i]) : "v"(fb), "v"(fc));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pb));
                if (OP == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
                if (OP == 7) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b));
                if (OP == 8) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (OP == 9) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x80" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));
                if (OP == 12) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 14) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 15) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(b), "v"(c) : "vcc");
                if (OP == 16) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(q[i]));
                if (OP == 17) asm volatile("v_min3_i32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 18) asm volatile("v_bfm_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 19) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (OP == 20) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(f[i]));
                if (OP == 21) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 22) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 23) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 24) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 25) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fb), "v"(fc));
                if (OP == 26) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (OP == 27) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(f[i]) : "v"(fb));
                if (OP == 28) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
                if (OP == 30) asm volatile("v_min_f32 %0, %1, %0" : "+v"(f[i]) : "v"(fb));
                if (OP == 31) asm volatile("v_sub_u32 %0, %1, %0 clamp" : "+v"(a[i]) : "v"(b));
                if (OP == 32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
                if (OP == 33) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 34) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 35) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
                if (OP == 36) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (OP == 37) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(fb) : "vcc");
                if (OP == 38) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 clamp" : "+v"(p[i]) : "v"(pb), "v"(pb));
                if (OP == 39) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)f[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32) + (uint32_t)p[i].x + (uint32_t)p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int waves_per_simd) {
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
    const int iters = 10000;
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
    printf("%-18s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms, ms * 1e6 / inst_per_simd,
           ms * 1e6 / inst_per_simd * 2.4);
    hipFree(out);
}

// Latency of dependent LDS accesses: one lane per wavefront chases a pointer chain through LDS (what a merge of the GPU triangulation does)
template <int WIDTH>
__global__ __launch_bounds__(256) void chase(uint32_t *out, int steps) {
    __shared__ uint32_t lds[4096 * 4];
    for (int i = threadIdx.x; i < 4096 * 4; i += blockDim.x) lds[i] = ((i / 4) * 1237 + 17) % 4096;  // word 0 of 16-byte record r: next record
    __syncthreads();
    uint32_t p = threadIdx.x % 4096, acc = 0;
    if ((threadIdx.x & 63) == 0) {
        for (int s = 0; s < steps; s++) {
            if (WIDTH == 1) {
                p = lds[p * 4];
            } else if (WIDTH == 3) {  // 12-byte read
                const uint32_t x = lds[p * 4], y = lds[p * 4 + 1], z = lds[p * 4 + 2];
                acc += y ^ z;
                p = x;
            } else {  // two dependent narrow reads per step (what 16-bit field accesses cost)
                const uint32_t x = ((volatile uint16_t *)lds)[p * 8];
                const uint32_t y = ((volatile uint16_t *)lds)[(x % 4096) * 8 + 2];
                acc += y;
                p = x;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p + acc;
}
template <int WIDTH>
void run_chase(const char *name) {
    uint32_t *out;
    hipMalloc(&out, 256 * 256 * 4);
    const int steps = 200000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    chase<WIDTH><<<128, 256>>>(out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chase<WIDTH><<<128, 256>>>(out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.1f ns per step\n", name, ms * 1e6 / steps);
    hipFree(out);
}
int main() {
    static const char *names[40] = {"v_add_f32", "v_mul_f32", "v_max_f32", "v_fma_f32 clamp", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_cndmask_b32", "v_cmp_lt_i32",
                                    "v_ffbl_b32", "v_bitop3_b32", "v_lshl_add_u32", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_min3_i32",
                                    "v_bfm_b32", "v_rcp_f32", "v_cvt_f32_i32", "v_sub_u32", "v_xor_b32", "v_bfi_b32", "v_mad_i32_i24", "v_med3_f32", "v_lshl_add_u64", "v_sub_f32",
                                    "v_and_or_b32", "v_add_co_u32", "v_min_f32", "v_sub_u32 clamp", "v_fmac_f32", "v_mul_u32_u24", "v_alignbit_b32", "v_lshrrev_b32", "v_or_b32",
                                    "v_cmp_lt_f32", "v_pk_fma_f32 clamp", "v_ashrrev_i32"};
    for (int w : {2, 4}) {
#define R(i) run<i>(names[i], w);
        R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16) R(17) R(18) R(19) R(20) R(21) R(22) R(23) R(24) R(25) R(26) R(27) R(28) R(29)
        R(30) R(31) R(32) R(33) R(34) R(35) R(36) R(37) R(38) R(39)
    }
    run_chase<1>("LDS chase, ds_read_b32");
    run_chase<3>("LDS chase, 12-byte record");
    run_chase<2>("LDS chase, 2 dependent u16");
    return 0;
}
