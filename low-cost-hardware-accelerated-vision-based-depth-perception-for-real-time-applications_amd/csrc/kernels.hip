// gfx950 (MI355X / CDNA4) kernels of the ELAS stereo hot path.
//
// Every kernel is written for 64-wide wavefronts and compiled with -ffp-contract=off: the reference's
// serial path is plain IEEE arithmetic without fused multiply-add, and the disparity maps
// must match it bit for bit.  Reference citations are relative to /root/reference/src.
//
// Integer byte work (descriptors, SAD) uses v_sad_u8 on 4-byte words; there is no matrix contraction here and
// therefore no MFMA.  Batch dimension: blockIdx.z (or .y) walks the pairs of a launch.
#include "sv_kernels.h"

#include <algorithm>
#include <atomic>

namespace sv {

thread_local LaunchHook g_launch_hook = {nullptr, nullptr};

const char *kernel_name(int id) {
    static const char *names[K_COUNT] = {"descriptor", "support_match", "support_filter", "grid_mark", "grid_dilate", "plane_fit", "triangles_raster", "dense_match", "lr_check",
                                         "delaunay_gpu", "ccl_band", "ccl_finish", "gap_rows", "gap_cols", "adaptive_mean", "median", "output"};
    return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}


// ------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }

__device__ __forceinline__ uint32_t sad16(const uint4 &a, const uint4 &b) {
    uint32_t s = sad4(a.x, b.x, 0u);
    s = sad4(a.y, b.y, s);
    s = sad4(a.z, b.z, s);
    return sad4(a.w, b.w, s);
}

__device__ __forceinline__ uint32_t sad16(const uint4 &a, const uint4 &b, uint32_t s) {  // continues an accumulation
    s = sad4(a.x, b.x, s);
    s = sad4(a.y, b.y, s);
    s = sad4(a.z, b.z, s);
    return sad4(a.w, b.w, s);
}

// acc + (SAD of the 16 bytes << 16): v_sad_hi_u8 adds its sum to the HIGH half, so a chain started with acc = tie-break bits builds
// the comparison key (energy << 16 | tie-break) without a shift; acc may carry a (negative) additive energy term in its high half
__device__ __forceinline__ int sad16_key(const uint4 &a, const uint4 &b, int acc) {
    uint32_t s = __builtin_amdgcn_sad_hi_u8(a.x, b.x, (uint32_t)acc);
    s = __builtin_amdgcn_sad_hi_u8(a.y, b.y, s);
    s = __builtin_amdgcn_sad_hi_u8(a.z, b.z, s);
    return (int)__builtin_amdgcn_sad_hi_u8(a.w, b.w, s);
}

// sum |byte - 128| over the 16 descriptor bytes (elas.cpp:296-298, 732-734)
__device__ __forceinline__ uint32_t texture16(const uint4 &a) {
    const uint4 mid = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
    return sad16(a, mid);
}

__device__ __forceinline__ uint4 ld16(const uint8_t *p) { return *reinterpret_cast<const uint4 *>(p); }

// Element `idx` of a map whose base pointer is uniform over the workgroup: a 32-bit BYTE offset beside the base (a map is far below
// 4 GB) lets the compiler address with the scalar base + one 32-bit VGPR (global_load ... v, s[base]) instead of building a 64-bit
// address per access with v_mad_u64_u32 / v_lshl_add_u64.  Used where the SQ_INSTS_VALU pass showed fewer instructions (tools/pmc_ab.sh:
// adaptive mean -4 %, median -4 %, L/R check -9 %, gap rows, Sobel, raster -2 %); the matching kernels' staging loads, the speckle bands and
// the gap columns issued MORE with it (+1 - 8 %) and keep their 64-bit arithmetic.
template <class T>
__device__ __forceinline__ T map_ld(const T *base, uint32_t idx) { return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + idx * (uint32_t)sizeof(T)); }
template <class T>
__device__ __forceinline__ void map_st(T *base, uint32_t idx, T v) { *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + idx * (uint32_t)sizeof(T)) = v; }


// work counters (sv_debug_counters): one atomic per wavefront, only in the COUNT instantiations of the matching kernels
__device__ __forceinline__ void count_add(unsigned long long *slot, int mine) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(slot, (unsigned long long)mine);
}

__device__ __forceinline__ int sat_u8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

// ------------------------------------------------------------------------------------------------------------
// K1  Sobel gradients; 16-byte descriptors are never stored
//     reference: common_includes/elas/filter.cpp:416-424 (sobel3x3 = :380-413 + :235-275 + :183-229)
//                common_includes/elas/descriptor.cpp:96-124
//     The reference materialises one 16-byte descriptor per pixel (12 du + 4 dv samples around it).  Here only the two
//     gradient images are stored (2 bytes per pixel instead of 16), as byte planes with padded rows, and the matching
//     kernels assemble the descriptors they stage in LDS from them (expand_quad): 32 B/px of descriptor stores and
//     ~45 B/px of descriptor re-reads per pair become 4 B/px of stores and plane reads that mostly hit in L2 / MALL.
//     Pixels outside [3,W-3)x[3,H-3) get the canonical zero descriptor (the reference leaves them uninitialised).
// ------------------------------------------------------------------------------------------------------------
// Plane geometry: [pair][image][du | dv][H][P] bytes, image column x at byte GRAD_MARGIN + x of its row; P is a multiple of
// 16 and leaves >= 16 bytes on either side, so 4-byte words at columns c-4, c, c+4 (c a multiple of 4, 0 <= c < W+4) are
// aligned and in bounds.  The margins are never written (zero from the allocation) and never reach a valid descriptor.
#define GRAD_MARGIN 16
__host__ __device__ inline int grad_pitch(int W) { return ((W + 15) & ~15) + 2 * GRAD_MARGIN; }
__host__ __device__ inline size_t grad_plane_bytes(const Dims &d) { return (size_t)d.H * grad_pitch(d.W); }

// LDS tile of k_sobel: gray words, tile column word cw <-> image columns x0 - 4 + 4*cw .. +3
#define SOB_TW 256                 // output tile width: one wavefront = one row of 64 words
#define SOB_TH 16                  // output tile height: four rows per wavefront
#define SOB_GW (SOB_TW / 4 + 2)    // gray words per tile row: columns x0-4 .. x0+SOB_TW+3; rows y0-1 .. y0+SOB_TH

struct __attribute__((packed)) UnalignedWord {  // caller's image rows have any alignment (stride 1242): one unaligned dword load
    uint32_t v;
};

struct __attribute__((aligned(4))) DwordQuad {  // four dwords at any dword-aligned address: one global_load_dwordx4
    uint32_t x, y, z, w;
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
// _mm_packus_epi16 for one pair: two int16 saturated to [0, 255], packed into bytes 0 and 1
__device__ __forceinline__ uint32_t sat_pk_u8(s2 x) {
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, x)));
    return r;  // (callers take bytes 0 and 1)
}

__global__ __launch_bounds__(256) void k_sobel(const uint8_t *__restrict__ left, const uint8_t *__restrict__ right, size_t in_pair_stride, int stride,
                                               uint8_t *__restrict__ grad, Dims d) {
    const int img = blockIdx.z & 1, pair = blockIdx.z >> 1;
    const uint8_t *I = (img ? right : left) + (size_t)pair * in_pair_stride;
    const int P = grad_pitch(d.W);
    uint8_t *DU = grad + ((size_t)(pair * 2 + img) * 2) * grad_plane_bytes(d) + GRAD_MARGIN, *DV = DU + grad_plane_bytes(d);
    const int x0 = blockIdx.x * SOB_TW, y0 = blockIdx.y * SOB_TH;
    __shared__ uint32_t g[(SOB_TH + 2) * SOB_GW];
    const int tid = threadIdx.x;
    // gray tile, one word (4 pixels) per item; pixels outside the image read as 0 (they never reach a valid descriptor)
    {  // a thread owns one word column of the tile and every third row: the column tests once, all its loads in flight together
        constexpr int RPP = 256 / SOB_GW, NLD = (SOB_TH + 2 + RPP - 1) / RPP;  // 3 rows per pass, 6 passes
        const int cw = tid % SOB_GW, rp = tid / SOB_GW;
        const int xb = x0 - 4 + 4 * cw;
        const bool live = rp < RPP, whole = xb >= 0 && xb + 3 < d.W;
        uint32_t w[NLD];
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP, y = y0 - 1 + r;
            w[t] = 0;
            if (live && r < SOB_TH + 2 && y >= 0 && y < d.H) {
                const uint32_t q = (uint32_t)(y * stride + xb);
                if (whole) {
                    w[t] = reinterpret_cast<const UnalignedWord *>(reinterpret_cast<const char *>(I) + q)->v;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int x = xb + j;
                        if (x >= 0 && x < d.W) w[t] |= (uint32_t)map_ld(I, q + (uint32_t)j) << (8 * j);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP;
            if (live && r < SOB_TH + 2) g[r * SOB_GW + cw] = w[t];
        }
    }
    __syncthreads();
    // one thread = 4 pixels (one word) of a row: gray rows r, r+1, r+2 of the tile (y-1, y, y+1), gray columns c-1 .. c+4
    const int wq = tid & 63;
#pragma unroll
    for (int kk = 0; kk < SOB_TH / 4; kk++) {
        const int r = (tid >> 6) + 4 * kk;
        const int y = y0 + r, x = x0 + 4 * wq;
        uint32_t a[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const uint32_t *gr = &g[(r + rr) * SOB_GW + wq];
            a[rr][0] = gr[0];
            a[rr][1] = gr[1];
            a[rr][2] = gr[2];
        }
        // Two 16-bit lanes per register, as the reference's SSE2 code has eight (filter.cpp:183-275): the gray columns c-1 .. c+4 of
        // a row as the pairs (c-1, c), (c+1, c+2), (c+3, c+4); S = (1,2,1) and T = (1,0,-1) down the three rows; then
        //   du(c+j) = ((S[j] - S[j+2]) >> 2) + 128        -> (du0, du1) = P0 - P1, (du2, du3) = P1 - P2 lane by lane
        //   dv(c+j) = ((T[j] + 2 T[j+1] + T[j+2]) >> 2) + 128   with the odd pairs (c, c+1), (c+2, c+3) cut out by v_perm_b32
        // and packus = v_sat_pk_u8_i16.  46 VALU instructions per four pixels (100 with one 32-bit lane per value).
        us2 Pc[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            Pc[rr][0] = __builtin_bit_cast(us2, __builtin_amdgcn_perm(a[rr][1], a[rr][0], 0x0C040C03u));
            Pc[rr][1] = __builtin_bit_cast(us2, __builtin_amdgcn_perm(a[rr][1], a[rr][1], 0x0C020C01u));
            Pc[rr][2] = __builtin_bit_cast(us2, __builtin_amdgcn_perm(a[rr][2], a[rr][1], 0x0C040C03u));
        }
        s2 S[3], T[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            S[q] = __builtin_bit_cast(s2, (us2)(Pc[0][q] + Pc[2][q] + (us2)(Pc[1][q] << 1)));  // <= 1020
            T[q] = __builtin_bit_cast(s2, Pc[0][q]) - __builtin_bit_cast(s2, Pc[2][q]);
        }
        const s2 Q0 = __builtin_bit_cast(s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, T[1]), __builtin_bit_cast(uint32_t, T[0]), 0x05040302u));  // T of (c, c+1)
        const s2 Q1 = __builtin_bit_cast(s2, __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, T[2]), __builtin_bit_cast(uint32_t, T[1]), 0x05040302u));  // T of (c+2, c+3)
        const s2 k128 = {128, 128};
        const uint32_t du01 = sat_pk_u8(((S[0] - S[1]) >> 2) + k128), du23 = sat_pk_u8(((S[1] - S[2]) >> 2) + k128);
        const uint32_t dv01 = sat_pk_u8(((T[0] + (s2)(Q0 << 1) + T[1]) >> 2) + k128), dv23 = sat_pk_u8(((T[1] + (s2)(Q1 << 1) + T[2]) >> 2) + k128);
        const uint32_t du_w = __builtin_amdgcn_perm(du23, du01, 0x05040100u), dv_w = __builtin_amdgcn_perm(dv23, dv01, 0x05040100u);
        if (y < d.H && x < ((d.W + 3) & ~3)) {  // whole words: the bytes beyond W land in the row's right margin and are never used
            const uint32_t q = (uint32_t)(y * P + x);  // (byte offset inside the plane; the plane pointers are uniform)
            *reinterpret_cast<uint32_t *>(DU + q) = du_w;
            *reinterpret_cast<uint32_t *>(DV + q) = dv_w;
        }
    }
}

size_t grad_bytes_per_pair(const KParams &k) { return 4 * grad_plane_bytes(k.d); }

void launch_sobel(const KParams &k, const uint8_t *left, const uint8_t *right, size_t in_pair_stride, int stride, const SlotDev &s, int n, hipStream_t st) {
    dim3 grid((k.d.W + SOB_TW - 1) / SOB_TW, (k.d.H + SOB_TH - 1) / SOB_TH, n * 2);
    SV_LAUNCH(K_DESCRIPTOR, k_sobel, grid, dim3(256), 0, st, left, right, in_pair_stride, stride, s.grad, k.d);
}

// byte i (0..7) of the 8-byte little-endian pair (lo, hi)
__device__ __forceinline__ uint32_t byte_of(uint32_t lo, uint32_t hi, int i) { return i < 4 ? (lo >> (8 * i)) & 0xFFu : (hi >> (8 * (i - 4))) & 0xFFu; }

// The 16 descriptor bytes of pixel column c+J of a quad (descriptor.cpp:105-121) from the quad's du / dv words.  Each source
// row is a 12-byte window (a, b, c) = columns c-4 .. c+7; the bytes one output word needs always lie inside 8 consecutive
// window bytes, so a word is one or two v_perm_b32 (selector byte 0-3: low operand, 4-7: high operand, 0x0C: zero).
struct DescWin {
    uint32_t r0, r4;             // du rows y-2, y+2: columns c..c+3
    uint32_t r1a, r1b, r1c;      // du row y-1
    uint32_t r2a, r2b, r2c;      // du row y
    uint32_t r3a, r3b, r3c;      // du row y+1
    uint32_t v1, v3;             // dv rows y-1, y+1: columns c..c+3
    uint32_t v2a, v2b, v2c;      // dv row y
};

template <int J>
__device__ __forceinline__ uint4 desc_assemble(const DescWin &w) {
    // triple (x-2, x, x+2): window bytes 2+J, 4+J, 6+J
    constexpr bool t_hi = 6 + J >= 8;
    constexpr uint32_t tb = t_hi ? 4 : 0, t0 = 2 + J - tb, t1 = 4 + J - tb, t2 = 6 + J - tb;
    const uint32_t row1 = __builtin_amdgcn_perm(t_hi ? w.r1c : w.r1b, t_hi ? w.r1b : w.r1a, 0x0Cu | (t0 << 8) | (t1 << 16) | (t2 << 24));
    const uint32_t row3 = __builtin_amdgcn_perm(t_hi ? w.r3c : w.r3b, t_hi ? w.r3b : w.r3a, t0 | (t1 << 8) | (t2 << 16) | (0x0Cu << 24));
    // (x-1, x, x, x+1): window bytes 3+J, 4+J, 4+J, 5+J
    constexpr bool m_hi = 5 + J >= 8;
    constexpr uint32_t mb = m_hi ? 4 : 0, m0 = 3 + J - mb, m1 = 4 + J - mb, m2 = 5 + J - mb;
    uint4 o;
    o.x = __builtin_amdgcn_perm(row1, w.r0, (uint32_t)J | (5u << 8) | (6u << 16) | (7u << 24));                     // du (0,-2) (-2,-1) (0,-1) (2,-1)
    o.y = __builtin_amdgcn_perm(m_hi ? w.r2c : w.r2b, m_hi ? w.r2b : w.r2a, m0 | (m1 << 8) | (m1 << 16) | (m2 << 24));  // du (-1,0) (0,0) (0,0) (1,0)
    o.z = __builtin_amdgcn_perm(row3, w.r4, 4u | (5u << 8) | (6u << 16) | ((uint32_t)J << 24));                     // du (-2,1) (0,1) (2,1) (0,2)
    const uint32_t mid = __builtin_amdgcn_perm(m_hi ? w.v2c : w.v2b, m_hi ? w.v2b : w.v2a, 0x0Cu | (m0 << 8) | (m2 << 16) | (0x0Cu << 24));
    const uint32_t lowr = __builtin_amdgcn_perm(mid, w.v1, (uint32_t)J | (5u << 8) | (6u << 16) | (0x0Cu << 24));
    o.w = __builtin_amdgcn_perm(lowr, w.v3, 4u | (5u << 8) | (6u << 16) | ((uint32_t)J << 24));                      // dv (0,-1) (-1,0) (1,0) (0,1)
    return o;
}

// Gradient planes of one image (k_sobel): DU / DV point at (row 0, column 0)
struct GradImg {
    const uint8_t *DU, *DV;
    int P;
};

__device__ __forceinline__ GradImg grad_image(const uint8_t *grad, const Dims &d, int pair, int img) {
    GradImg g;
    g.P = grad_pitch(d.W);
    g.DU = grad + ((size_t)(pair * 2 + img) * 2) * grad_plane_bytes(d) + GRAD_MARGIN;
    g.DV = g.DU + grad_plane_bytes(d);
    return g;
}

// descriptor.cpp:48-50, 96-98: rows that carry descriptors (every second line from 4 at half resolution)
__device__ __forceinline__ bool desc_row_ok(const Dims &d, int y) { return y < d.H - 3 && (d.sub ? (y >= 4 && !(y & 1)) : y >= 3); }

// The four descriptors of image columns c .. c+3 (c a multiple of 4) of row y -> out[0..3].  16 aligned word loads from the
// planes (consecutive lanes = consecutive quads: every load instruction is one contiguous 256-byte piece), 32 v_perm_b32.
__device__ __forceinline__ void expand_quad(const GradImg &g, const Dims &d, int y, int c, uint4 *out) {
    if (!desc_row_ok(d, y)) {  // wave-uniform in every caller (one row per workgroup / loop trip)
        out[0] = out[1] = out[2] = out[3] = make_uint4(0, 0, 0, 0);
        return;
    }
#define LDW(base, row, off) (*reinterpret_cast<const uint32_t *>((base) + (size_t)(row) * g.P + c + (off)))
    DescWin w;
    w.r0 = LDW(g.DU, y - 2, 0), w.r4 = LDW(g.DU, y + 2, 0);
    w.r1a = LDW(g.DU, y - 1, -4), w.r1b = LDW(g.DU, y - 1, 0), w.r1c = LDW(g.DU, y - 1, 4);
    w.r2a = LDW(g.DU, y, -4), w.r2b = LDW(g.DU, y, 0), w.r2c = LDW(g.DU, y, 4);
    w.r3a = LDW(g.DU, y + 1, -4), w.r3b = LDW(g.DU, y + 1, 0), w.r3c = LDW(g.DU, y + 1, 4);
    w.v1 = LDW(g.DV, y - 1, 0), w.v3 = LDW(g.DV, y + 1, 0);
    w.v2a = LDW(g.DV, y, -4), w.v2b = LDW(g.DV, y, 0), w.v2c = LDW(g.DV, y, 4);
#undef LDW
    const uint4 z = make_uint4(0, 0, 0, 0);
    out[0] = (c + 0 >= 3 && c + 0 < d.W - 3) ? desc_assemble<0>(w) : z;
    out[1] = (c + 1 >= 3 && c + 1 < d.W - 3) ? desc_assemble<1>(w) : z;
    out[2] = (c + 2 >= 3 && c + 2 < d.W - 3) ? desc_assemble<2>(w) : z;
    out[3] = (c + 3 >= 3 && c + 3 < d.W - 3) ? desc_assemble<3>(w) : z;
}

// sum |byte - 128| over the 16 bytes of the descriptor at (x, y), a valid descriptor position (elas.cpp:296-298): straight from
// the planes, byte by byte (the lattice points of the support search: a few thousand per pair)
__device__ __forceinline__ uint32_t texture_at(const GradImg &g, int y, int x) {
    auto a = [&](const uint8_t *base, int dx, int dy) { return (uint32_t)abs((int)base[(size_t)(y + dy) * g.P + x + dx] - 128); };
    return a(g.DU, 0, -2) + a(g.DU, -2, -1) + a(g.DU, 0, -1) + a(g.DU, 2, -1) + a(g.DU, -1, 0) + 2u * a(g.DU, 0, 0) + a(g.DU, 1, 0) + a(g.DU, -2, 1) + a(g.DU, 0, 1) +
           a(g.DU, 2, 1) + a(g.DU, 0, 2) + a(g.DV, 0, -1) + a(g.DV, -1, 0) + a(g.DV, 1, 0) + a(g.DV, 0, 1);
}

// Debug configuration only: the descriptor images the reference would hold, for the stage snapshot (same expand_quad)
__global__ __launch_bounds__(256) void k_expand_all(const uint8_t *__restrict__ grad, uint8_t *__restrict__ desc, Dims d) {
    const int img = blockIdx.z & 1, pair = blockIdx.z >> 1;
    const int y = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= d.W) return;
    const GradImg g = grad_image(grad, d, pair, img);
    uint4 o[4];
    expand_quad(g, d, y, 4 * q, o);
    uint4 *out = reinterpret_cast<uint4 *>(desc + ((size_t)(pair * 2 + img) * d.N + (size_t)y * d.W) * 16);
    for (int j = 0; j < 4; j++)
        if (4 * q + j < d.W) out[4 * q + j] = o[j];
}

// A small block of words between page-locked host memory and HBM, moved by a kernel on the stream of the kernels that produce / consume
// it (latency mode only: single pairs).  The runtime's hipMemcpyAsync puts such a copy on an SDMA engine or on a blit kernel in a queue of
// its own, and the dependency between that and the stream's kernels costs 10 - 12 us each way (tools/latency_trace.py under rocprofv3:
// k_support -> 12 us -> copy; copy -> 12 us -> k_planes); a kernel in the same queue starts when its predecessor ends.  Streamed batches
// keep the DMA engines: there the copies overlap other chunks' kernels and the CUs have better things to do.
__global__ __launch_bounds__(256) void k_copy_block(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16, size_t tail_bytes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail_bytes) reinterpret_cast<uint8_t *>(dst + n16)[threadIdx.x] = reinterpret_cast<const uint8_t *>(src + n16)[threadIdx.x];
}

void launch_copy_block(void *dst, const void *src, size_t bytes, hipStream_t st) {  // both 16-byte aligned, device-visible
    if (!bytes) return;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) throw std::invalid_argument("launch_copy_block: unaligned block");
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_copy_block, dim3((unsigned)std::max<size_t>(1, (n16 + 255) / 256)), dim3(256), 0, st, static_cast<uint4 *>(dst), static_cast<const uint4 *>(src), n16, bytes - n16 * 16);
}

void launch_expand_debug(const KParams &k, const SlotDev &s, int n, uint8_t *desc, hipStream_t st) {
    hipLaunchKernelGGL(k_expand_all, dim3((k.d.W / 4 + 256) / 256, k.d.H, n * 2), dim3(256), 0, st, s.grad, desc, k.d);
}

// ------------------------------------------------------------------------------------------------------------
// K2  support matching on the lattice: the only full-range disparity search
//     reference: serial_includes/elas/elas.cpp:266-371 (computeMatchingDisparity), :387-411 (loop + L/R check)
//     energy = 4-corner SAD (64 bytes); (best energy, lowest best d) and the second order statistic of the energies.
// ------------------------------------------------------------------------------------------------------------
// Rows v-2 and v+2 of both descriptor images are staged in LDS for a run of 64 consecutive lattice points of one lattice
// row: the forward search reads the right image over [u-2-dmax, u+2], the backward check the left image over
// [u-d-2, u-d+2+dmax]; neighbouring lattice points (5 px apart) share almost all of it.
//
// One LANE per lattice point and quarter of the disparity range (a workgroup = 64 consecutive lattice points x 4 wavefronts,
// one per quarter), disparities scanned in ascending order like the reference's loop: no cross-lane reduction, only a merge of
// four (best, runner-up) records per point through LDS.  The descriptor at u-d+2 is the one loaded four steps earlier for
// u-(d-4)-2, so a register rotation (two sets of four, ping-ponged by an eight-step loop body) halves the LDS reads:
// 2 x ds_read_b128 + 16 x v_sad_hi_u8 (one chain that starts from d and leaves energy << 16 | d) + 2 bookkeeping
// instructions per disparity.  Consecutive lanes are `step` descriptors (80 bytes at step 5) apart: conflict-free for
// 128-bit LDS reads.  (One wavefront per lattice point with lanes over d, the first version, spent as many instructions on
// the 64-lane reduction as on the SADs: 8.5 us per pair against 4.1.)
#define SUP_THREADS 256
#define SUP_POINTS 64
#define SUP_SPLIT (SUP_THREADS / SUP_POINTS)
#define SUP_SPLIT_ALONE 8

struct SupRows {            // one staged image: rows v-2 and v+2, columns [c0, c0+n)
    const uint4 *r0, *r1;
    int c0;
};

// (best key, runner-up key) of the keys seen so far, key = energy<<16 | d: the smallest key is the lowest energy at the lowest d
// (first-wins under the reference's strict <, elas.cpp:352-360), the energy of the second smallest key is the second order
// statistic of the energy multiset, which is what min_2_E ends as.
__device__ __forceinline__ void top2_push(uint32_t &k1, uint32_t &k2, uint32_t key) {
    uint32_t m;  // k1 <= k2: the median of the three is the new runner-up
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k1), "v"(k2), "v"(key));
    k2 = m;
    k1 = min(k1, key);
}

constexpr uint32_t SUP_KEY_NONE = 0x7FFFFFFFu;  // energy 32767, d 0xFFFF: the reference's initial min_1_E / min_2_E

// energies of the disparities [d_lo, d_hi] of the lattice point at column u of image A against image B (elas.cpp:341-360)
template <bool RIGHT>
__device__ __forceinline__ void support_scan(const SupRows &A, const SupRows &B, int u, int d_lo, int d_hi, uint32_t &k1, uint32_t &k2) {
    const int ua = u - A.c0;
    const uint4 a0 = A.r0[ua - 2], a1 = A.r0[ua + 2], a2 = A.r1[ua - 2], a3 = A.r1[ua + 2];
    // P(d) = B[u + dir*(d+2)] is loaded at step d; Q(d) = B[u + dir*(d-2)] = P(d-4) comes from the rotation
    constexpr int dir = RIGHT ? 1 : -1;
    const uint4 *b0 = B.r0 + (u - B.c0), *b1 = B.r1 + (u - B.c0);
    uint4 p0[4], p1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p0[j] = b0[dir * (d_lo + j - 2)];
        p1[j] = b1[dir * (d_lo + j - 2)];
    }
    // step j of a block of four at disparity d + off + j: the descriptors the block before it loaded (o0/o1) are the rotated
    // operands, the fresh ones go to n0/n1.  One accumulation chain, the older operands first: the loads have longer to land.
#define SUP_STEP(j, off, o0, o1, n0, n1)                                                                                    \
    {                                                                                                                       \
        const uint4 q0 = o0[j], q1 = o1[j];                                                                                 \
        n0[j] = b0[dir * (d + (off) + (j) + 2)];                                                                            \
        n1[j] = b1[dir * (d + (off) + (j) + 2)];                                                                            \
        const int dk = d + (off) + (j); /* the chain starts from d and adds the SADs to the high half: energy << 16 | d */   \
        const int key = RIGHT ? sad16_key(a3, n1[j], sad16_key(a1, n0[j], sad16_key(a2, q1, sad16_key(a0, q0, dk))))        \
                              : sad16_key(a2, n1[j], sad16_key(a0, n0[j], sad16_key(a3, q1, sad16_key(a1, q0, dk))));       \
        top2_push(k1, k2, (uint32_t)key);                                                                                   \
    }
    uint4 r0[4], r1[4];
    int d = d_lo;
    for (; d + 7 <= d_hi; d += 8) {  // two blocks per trip, ping-ponging between the register sets: no copies
        SUP_STEP(0, 0, p0, p1, r0, r1) SUP_STEP(1, 0, p0, p1, r0, r1) SUP_STEP(2, 0, p0, p1, r0, r1) SUP_STEP(3, 0, p0, p1, r0, r1)
        SUP_STEP(0, 4, r0, r1, p0, p1) SUP_STEP(1, 4, r0, r1, p0, p1) SUP_STEP(2, 4, r0, r1, p0, p1) SUP_STEP(3, 4, r0, r1, p0, p1)
    }
    if (d + 3 <= d_hi) {
        SUP_STEP(0, 0, p0, p1, r0, r1) SUP_STEP(1, 0, p0, p1, r0, r1) SUP_STEP(2, 0, p0, p1, r0, r1) SUP_STEP(3, 0, p0, p1, r0, r1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p0[j] = r0[j];
            p1[j] = r1[j];
        }
        d += 4;
    }
    if (d <= d_hi) SUP_STEP(0, 0, p0, p1, p0, p1)
    if (d + 1 <= d_hi) SUP_STEP(1, 0, p0, p1, p0, p1)
    if (d + 2 <= d_hi) SUP_STEP(2, 0, p0, p1, p0, p1)
#undef SUP_STEP
}

// the four partial records of a point -> (best, runner-up) of the whole range
template <int SPLIT>
__device__ __forceinline__ uint2 support_merge(const uint2 *rec, int point) {
    uint2 m = rec[point];
#pragma unroll
    for (int g = 1; g < SPLIT; ++g) {
        const uint2 o = rec[g * SUP_POINTS + point];
        m.y = min(min(m.y, o.y), max(m.x, o.x));
        m.x = min(m.x, o.x);
    }
    return m;
}

// elas.cpp:279 (border), :318-326 (range [max(disp_min, 0), disp_max_valid], at least 11 disparities): the highest disparity to scan,
// -1 = no match possible
template <bool RIGHT>
__device__ __forceinline__ int support_range(const Dims &d, int u, int v) {
    if (!(u >= 5 && u <= d.W - 6 && v >= 5 && v <= d.H - 6)) return -1;
    const int dmax = RIGHT ? min(d.disp_max, d.W - u - 5) : min(d.disp_max, u - 5);
    return dmax - d.disp_min < 10 ? -1 : dmax;
}

// elas.cpp:296-300 (texture of the centre descriptor), :364 (ratio test)
__device__ __forceinline__ int support_decide(const KParams &k, uint32_t texture, uint2 m) {
    if ((int)texture < k.support_texture) return -1;
    const float E1 = (float)(m.x >> 16), E2 = (float)(m.y >> 16);
    return E1 < k.support_threshold * E2 ? (int)(m.x & 0xFFFFu) : -1;
}

// SPLIT wavefronts per workgroup, each with a SPLIT-th of the disparity range: SUP_SPLIT (4) inside the pipeline; SUP_SPLIT_ALONE (8) when the
// launch cannot fill the GPU anyway (a single pair: 290 workgroups on 256 CUs) - a lane's scan is then half as long.
// (The lattice of a single pair copied to the host by this kernel's last workgroup instead of by a copy kernel behind it - one launch
// boundary less - was measured: one workgroup moving 37 KB is five rounds of dependent accesses, 67 instead of 20 us of waiting.)
template <bool COUNT, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void k_support(KParams k, const uint8_t *__restrict__ grad, int16_t *__restrict__ dcan, unsigned long long *__restrict__ counters) {
    constexpr int SUP_THREADS_T = 64 * SPLIT;
    const Dims &d = k.d;
    extern __shared__ uint4 sup_lds[];
    const int pair = blockIdx.z, vc = blockIdx.y + 1;
    const int uc0 = 1 + blockIdx.x * SUP_POINTS, uc1 = min(uc0 + SUP_POINTS, d.Wc);  // candidates [uc0, uc1)
    const int v = vc * d.step;
    const GradImg g1 = grad_image(grad, d, pair, 0), g2 = grad_image(grad, d, pair, 1);
    const int u_lo = uc0 * d.step, u_hi = (uc1 - 1) * d.step;
    // staged column ranges, clipped to the image; they start on a multiple of 4: descriptors are assembled four columns at a time
    const int r_c0 = max(u_lo - 2 - d.disp_max, 0) & ~3, r_c1 = min(u_hi + 2, d.W - 1);
    const int l_c0 = max(u_lo - 2 - d.disp_max, 0) & ~3, l_c1 = min(u_hi + 2 + d.disp_max, d.W - 1);
    const int qR = (r_c1 - r_c0 + 4) >> 2, qL = (l_c1 - l_c0 + 4) >> 2;  // quads per staged row
    const int nR = 4 * qR, nL = 4 * qL;
    uint2 *rec = reinterpret_cast<uint2 *>(sup_lds);  // [2 passes][SUP_SPLIT][SUP_POINTS]
    uint4 *sR0 = sup_lds + 2 * SUP_THREADS_T * sizeof(uint2) / sizeof(uint4), *sR1 = sR0 + nR, *sL0 = sR1 + nR, *sL1 = sL0 + nL;
    // rows v-2 and v+2 of both descriptor images, assembled from the gradient planes (rows without descriptors come out as zeros)
    for (int i = threadIdx.x; i < 2 * (qR + qL); i += SUP_THREADS_T) {
        const int row = i & 1, q = i >> 1;  // consecutive lanes alternate between the two rows of one quad column
        if (q < qR)
            expand_quad(g2, d, row ? v + 2 : v - 2, r_c0 + 4 * q, (row ? sR1 : sR0) + 4 * q);
        else
            expand_quad(g1, d, row ? v + 2 : v - 2, l_c0 + 4 * (q - qR), (row ? sL1 : sL0) + 4 * (q - qR));
    }
    __syncthreads();
    const SupRows L{sL0, sL1, l_c0}, R{sR0, sR1, r_c0};
    const int point = threadIdx.x & (SUP_POINTS - 1), part = __builtin_amdgcn_readfirstlane(threadIdx.x / SUP_POINTS);  // wave-uniform
    const int uc = uc0 + point, u = uc * d.step;
    const int qlen = (d.disp_max - d.disp_min + SPLIT) / SPLIT, d_lo = d.disp_min + part * qlen;  // this wavefront's share of [disp_min, disp_max]
    // left -> right
    const int dmax1 = uc < uc1 ? support_range<false>(d, u, v) : -1;
    uint32_t texture = 0;  // of the centre descriptor (elas.cpp:296-300)
    __shared__ uint32_t s_tex[SUP_POINTS];  // one of the four wavefronts of a point fetches its sixteen bytes, the others read the sum after the barrier
    if (part == 0) {
        if (dmax1 >= 0) texture = texture_at(g1, v, u);  // used after the search: its latency hides behind it
        s_tex[point] = texture;
    }
    uint32_t k1 = SUP_KEY_NONE, k2 = SUP_KEY_NONE;
    if (d_lo <= dmax1) support_scan<false>(L, R, u, d_lo, min(dmax1, d_lo + qlen - 1), k1, k2);
    rec[threadIdx.x] = make_uint2(k1, k2);
    __syncthreads();
    texture = s_tex[point];
    const int dd = dmax1 >= 0 ? support_decide(k, texture, support_merge<SPLIT>(rec, point)) : -1;
    // right -> left from the match (elas.cpp:404-409)
    const int u2 = u - dd;
    const int dmax2 = dd >= 0 ? support_range<true>(d, u2, v) : -1;
    if (dmax2 >= 0 && part == 0) texture = texture_at(g2, v, u2);
    k1 = k2 = SUP_KEY_NONE;
    if (d_lo <= dmax2) support_scan<true>(R, L, u2, d_lo, min(dmax2, d_lo + qlen - 1), k1, k2);
    rec[SUP_THREADS_T + threadIdx.x] = make_uint2(k1, k2);
    if (COUNT)  // 64-byte energies evaluated by this lane: its share of [0, dmax] in both directions
        count_add(counters + CNT_SUPPORT_ENERGIES, (d_lo <= dmax1 ? min(dmax1, d_lo + qlen - 1) - d_lo + 1 : 0) + (d_lo <= dmax2 ? min(dmax2, d_lo + qlen - 1) - d_lo + 1 : 0));
    __syncthreads();
    if (part == 0 && uc < uc1) {
        const int d2v = dmax2 >= 0 ? support_decide(k, texture, support_merge<SPLIT>(rec + SUP_THREADS_T, point)) : -1;
        const int res = (d2v >= 0 && abs(dd - d2v) <= k.lr_threshold) ? dd : -1;
        int16_t *lat = dcan + (size_t)pair * d.Wc * d.Hc;
        lat[(size_t)uc * d.Hc + vc] = (int16_t)res;  // transposed: the host filters scan u outer / v inner
        // row 0 / column 0 of the reference's calloc'd lattice stay 0 (elas.cpp:387; they count as valid d=0 neighbours in the
        // filters): written here instead of by a memset in front of the kernel
        if (vc == 1) lat[(size_t)uc * d.Hc] = 0;
        if (uc == 1) lat[vc] = 0;
        if (uc == 1 && vc == 1) lat[0] = 0;
    }
}

void launch_support(const KParams &k, const SlotDev &s, int n, hipStream_t st) {
    if (k.d.Wc < 2 || k.d.Hc < 2) {  // no lattice point besides row 0 / column 0: nothing to match, the lattice is all zero
        (void)hipMemsetAsync(s.dcan, 0, sizeof(int16_t) * (size_t)n * k.d.Wc * k.d.Hc, st);
        return;
    }
    const int span = (SUP_POINTS - 1) * k.d.step;
    dim3 grid((k.d.Wc - 1 + SUP_POINTS - 1) / SUP_POINTS, k.d.Hc - 1, n);
    const bool alone = !s.counters && (size_t)grid.x * grid.y * grid.z <= 512;  // fewer than two workgroups per CU
    const int threads = 64 * (alone ? SUP_SPLIT_ALONE : SUP_SPLIT);
    // two rows per image; each staged range starts up to 3 columns early and ends on a whole quad
    const size_t shmem = 2 * threads * sizeof(uint2) + sizeof(uint4) * 2 * ((size_t)(span + k.d.disp_max + 5 + 6) + (size_t)(span + 2 * k.d.disp_max + 5 + 6));
    static std::atomic<size_t> granted[64], granted_c[64], granted_a[64];
    if (s.counters) {
        ensure_dynamic_lds(k_support<true, SUP_SPLIT>, shmem, granted_c, "support_match");
        SV_LAUNCH(K_SUPPORT, (k_support<true, SUP_SPLIT>), grid, dim3(threads), shmem, st, k, s.grad, s.dcan, s.counters);
        return;
    }
    if (alone) {
        ensure_dynamic_lds(k_support<false, SUP_SPLIT_ALONE>, shmem, granted_a, "support_match");
        SV_LAUNCH(K_SUPPORT, (k_support<false, SUP_SPLIT_ALONE>), grid, dim3(threads), shmem, st, k, s.grad, s.dcan, s.counters);
        return;
    }
    ensure_dynamic_lds(k_support<false, SUP_SPLIT>, shmem, granted, "support_match");  // large disparity ranges: more than the default dynamic LDS limit
    SV_LAUNCH(K_SUPPORT, (k_support<false, SUP_SPLIT>), grid, dim3(threads), shmem, st, k, s.grad, s.dcan, s.counters);
}

// ------------------------------------------------------------------------------------------------------------
// K2b support lattice -> support point list, on the GPU, for lattices of any size
//     reference: elas.cpp:152-176 (removeInconsistentSupportPoints), :178-233 (removeRedundantSupportPoints x2, :419-420),
//                :422-433 (collect, u outer / v inner), :235-264 (addCornerSupportPoints)
//     The reference's filters are sequential in-place scans.  The inconsistency pass is made parallel by classification:
//     when a lattice point is visited, the neighbours LATER in scan order are still unmodified, so
//        c_late  = consistent neighbours later-or-equal in scan order (exact at visit time)
//        c_all   = consistent neighbours in the original lattice (upper bound at visit time)
//     c_late >= min_support -> certainly kept; c_all < min_support -> certainly dropped; only the rest ("uncertain": 1-5 % of
//     the valid points of a real pair, a few hundred of a 4K lattice's 330 000) depend on the fate of earlier points: a few
//     parallel refinement rounds settle most of them, what remains is resolved one after the other, in scan order, by one
//     wavefront that counts the <= 60 earlier neighbours with a ballot.
//     The redundancy passes only couple points of one column / one row, and along a line the scan is a recurrence over the
//     kept-flags of the previous five points: everything that does not depend on the walk (is the point valid, does it have a
//     match among the five ORIGINAL later neighbours, which of the five earlier neighbours match) is computed for all points
//     in parallel into one byte per point; the walk itself is five dependent instructions per point on LDS bytes.
//     Six small launches, grid-wide except the resolve step (whose work is the few uncertain points); state in global
//     memory (16 bytes per lattice point), no size limit:
//       k_filter_classify    one thread per lattice point: the 121-neighbour counts; ordered list of each block's uncertain points
//       k_filter_resolve     one workgroup per pair: refinement rounds + sequential rest over the uncertain points only
//       k_filter_vertical    one workgroup per strip of 8 lattice columns: flag bytes in parallel, one lane per column walks them
//       k_filter_horizontal  one workgroup per strip of 8 lattice rows: the same along the rows; counts the points per collect block
//       k_filter_collect / k_filter_corners
//                            ordered compaction over blocks of 1024 lattice points (positions = sum of the predecessors'
//                            counts), nearest point per image corner as a 64-bit key minimum, corner points
//     Lattice layout: transposed, T[u*Hc + v] (scan order == index order).
// ------------------------------------------------------------------------------------------------------------
#define FST_NONE 0
#define FST_KEEP 1
#define FST_DROP 2
#define FST_UNC 3

#define FLT_THREADS 256   // lattice points per collect block (1 024 until round 5: a 16-wavefront workgroup waits for a CU with four free wavefront slots per SIMD - 12 x its own duration inside the pipeline)
#define RSV_THREADS 256   // threads of the resolve workgroup

// exclusive prefix sum over the workgroup (NT threads): wavefront scans by shuffles, one barrier pair for the wavefront totals
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int val, int *s_wave, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = val;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        incl += lane >= off ? o : 0;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        const int c = s_wave[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - val;
}

__host__ __device__ inline int filter_collect_blocks(int lat) { return (lat + FLT_THREADS - 1) / FLT_THREADS; }

// ---- (1) classification (elas.cpp:152-176), one thread per lattice point.  A block of 256 consecutive indices needs the
// contiguous index span [first - 5*Hc - 5, last + 5*Hc + 5] of the transposed lattice: staged in LDS once, then the 11x11 window
// (win <= 5, fully unrolled: independent reads, all in flight together) comes from there.  The block's uncertain points go,
// in index order, to its own segment useg[first ...] (count in ucnt): the resolve step never has to look at the lattice.
#define FCL_THREADS 256
// The lattice the later steps work on: one 32-bit word per point, value (int16) in the low half, state byte in bits 16-23; `FCS_PAD`
// words of padding in front of and behind a pair's lattice, so that the resolve step may fetch the eleven neighbours of a lattice
// column as three wide loads wherever the column lies.
#define FCS_PAD 16
__host__ __device__ inline size_t filter_cs_stride(int lat) { return (size_t)lat + 2 * FCS_PAD; }
__device__ __forceinline__ int cs_value(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int cs_state(uint32_t w) { return (int)((w >> 16) & 0xFFu); }

__global__ __launch_bounds__(FCL_THREADS) void k_filter_classify(Dims d, int win, int thr, int need, const int16_t *__restrict__ dcan, uint32_t *__restrict__ fcs,
                                                                 uint32_t *__restrict__ useg, int32_t *__restrict__ ucnt) {
    // LDS: the lattice columns this block's 256 points and their windows touch, each padded to P = Hc + 10 rows; everything that is
    // not a valid lattice value - invalid points, rows / columns outside the lattice - holds FCL_SENT, which is further than any
    // threshold from every disparity, so a neighbour costs a read, a subtract, one unsigned compare (|a - b| <= t  <=>
    // (unsigned)(b - a + t) <= 2t) and an add-with-carry: no bounds tests, no validity tests
    constexpr int FCL_SENT = 0x4000;
    extern __shared__ int16_t fcl_lds[];
    __shared__ int s_wcnt[FCL_THREADS / 64];
    const int pair = blockIdx.y, Hc = d.Hc, Wc = d.Wc, lat = Wc * Hc, P = Hc + 10;
    const int16_t *G = dcan + (size_t)pair * lat;
    const int first = blockIdx.x * FCL_THREADS, last = min(first + FCL_THREADS, lat) - 1;
    const int c0 = first / Hc - 5, ncol = last / Hc - first / Hc + 11;  // staged columns c0 .. c0 + ncol - 1
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < ncol; c += FCL_THREADS / 64) {
        const int u = c0 + c;
        const bool col_in = u >= 0 && u < Wc;
        for (int r = lane; r < P; r += 64) {
            const int v = r - 5;
            int x = FCL_SENT;
            if (col_in && v >= 0 && v < Hc) {
                x = G[(size_t)u * Hc + v];
                x = x < 0 ? FCL_SENT : x;
            }
            fcl_lds[c * P + r] = (int16_t)x;
        }
    }
    __syncthreads();
    const int idx = first + threadIdx.x;
    uint8_t state = FST_NONE;
    int dd = -1;
    if (idx < lat) {
        const int u = idx / Hc, v = idx - u * Hc;
        const int16_t *ctr = fcl_lds + (u - c0) * P + 5 + v;  // the point itself; neighbour (du, dv) at ctr[du * P + dv]
        const int x = ctr[0];
        if (x != FCL_SENT) {
            dd = x;
            const uint32_t lo = (uint32_t)(dd - thr), span = 2u * (uint32_t)thr;
            uint32_t c_late = 0, c_early = 0;
#pragma unroll
            for (int du = -5; du <= 5; du++) {
                if (du < -win || du > win) continue;  // (wave-uniform)
                const int16_t *col = ctr + du * P;
#pragma unroll
                for (int dv = -5; dv <= 5; dv++) {
                    if (dv < -win || dv > win) continue;
                    const uint32_t hit = ((uint32_t)(int)col[dv] - lo) <= span ? 1u : 0u;
                    if (du > 0 || (du == 0 && dv >= 0))
                        c_late += hit;
                    else
                        c_early += hit;
                }
            }
            state = (uint8_t)((int)c_late >= need ? FST_KEEP : ((int)(c_late + c_early) < need) ? FST_DROP : (FST_UNC | (c_late << 2)));
        }
        fcs[(size_t)pair * filter_cs_stride(lat) + FCS_PAD + idx] = (uint32_t)(uint16_t)dd | ((uint32_t)state << 16);
    }
    const bool unc = (state & 3) == FST_UNC;
    const unsigned long long m = __ballot(unc);
    if (lane == 0) s_wcnt[wave] = (int)__popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < FCL_THREADS / 64; w++) {
        base += w < wave ? s_wcnt[w] : 0;
        total += s_wcnt[w];
    }
    if (unc) useg[(size_t)pair * lat + first + base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)idx;
    if (threadIdx.x == 0) ucnt[(size_t)pair * gridDim.x + blockIdx.x] = total;
}

// ---- (2) the uncertain points: parallel refinement rounds, then the rest in scan order by one wavefront; the states it
// leaves behind are final (KEEP / DROP).  One workgroup per pair; its work is proportional to the number of classify blocks
// and of uncertain points, not to the lattice.  Uncertain point q (in scan order) is entry q - pref[b] of the segment of the
// classify block b with pref[b] <= q < pref[b + 1]: a binary search in the blocks' prefix sums (LDS) spreads the points evenly
// over the threads however they cluster in the lattice (they do: along depth edges).
#define RSV_MAX_BLOCKS 8192  // classify blocks per pair whose prefix sums fit the LDS table (lattices up to 2 M points)
__global__ __launch_bounds__(RSV_THREADS) void k_filter_resolve(Dims d, int win, int thr, int need, uint32_t *fcs, const uint32_t *__restrict__ useg,
                                                                const int32_t *__restrict__ ucnt, int nb, uint32_t *__restrict__ ulist, uint32_t *__restrict__ urest, int32_t *__restrict__ bcnt,
                                                                int nb2) {
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < nb2; i += RSV_THREADS) bcnt[(size_t)pair * nb2 + i] = 0;  // the collect blocks' point counts: the horizontal pass adds to them
    const int Hc = d.Hc, lat = d.Wc * Hc;
    uint32_t *cs = fcs + (size_t)pair * filter_cs_stride(lat) + FCS_PAD;  // cs[idx]: value | state << 16
    uint8_t *stb = reinterpret_cast<uint8_t *>(cs) + 2;                   // the state byte of point idx: stb[4 * idx]
    const uint32_t *seg = useg + (size_t)pair * lat;
    uint32_t *list = ulist + (size_t)pair * lat, *rest = urest + (size_t)pair * lat;
    __shared__ int s_wave[RSV_THREADS / 64];
    extern __shared__ int s_pref[];  // [nb + 1]
    const int32_t *cnt = ucnt + (size_t)pair * nb;
    const int per = (nb + RSV_THREADS - 1) / RSV_THREADS;
    const int b_lo = min(tid * per, nb), b_hi = min(b_lo + per, nb);
    int mine = 0;
    for (int b = b_lo; b < b_hi; b++) mine += cnt[b];
    int n_unc;
    int pos = block_exclusive_scan<RSV_THREADS>(mine, s_wave, &n_unc);
    if (n_unc == 0) return;
    for (int b = b_lo; b < b_hi; b++) {
        s_pref[b] = pos;
        pos += cnt[b];
    }
    if (tid == 0) s_pref[nb] = n_unc;
    __syncthreads();
    for (int q = tid; q < n_unc; q += RSV_THREADS) {  // the ordered list: one entry per thread and trip
        int lo = 0, hi = nb;                          // largest b with pref[b] <= q (empty blocks share a prefix: take the last)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pref[mid] <= q) lo = mid;
            else hi = mid;
        }
        list[q] = seg[(size_t)lo * FCL_THREADS + (q - s_pref[lo])];
    }
    __threadfence_block();
    __syncthreads();
    // refinement rounds over the uncertain points, fully parallel: earlier neighbours that are certainly kept count
    // for sure, earlier neighbours that are certainly dropped never count.  (A round only reads states of EARLIER points and
    // only turns UNC into KEEP/DROP; both decisions stay valid whatever the remaining UNC points become, so concurrent
    // updates are benign.)  The rounds end as soon as one of them leaves nothing behind.
    int left = 1;
    for (int round = 0; round < 3 && left; round++) {
        int open = 0;
        for (int q = tid; q < n_unc; q += RSV_THREADS) {
            const int idx = (int)list[q];
            const uint32_t w0 = __hip_atomic_load(&cs[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int s0 = cs_state(w0);
            if ((s0 & 3) != FST_UNC) continue;
            const int dd = cs_value(w0), c_late = s0 >> 2;
            const int u = idx / Hc, v = idx - u * Hc;
            // the 60 earlier neighbours: rows v-5 .. v+5 of the five columns before u and rows v-5 .. v-1 of column u, each column's
            // eleven words as three wide loads (plain loads: a word read while another thread of this workgroup settles that point
            // carries the old or the new state, and either is valid for a round); rows beyond the column's ends are masked below
            // (they are the neighbouring column's words, or the padding)
            uint32_t nw[6][12];
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const uint32_t *col = cs + (ptrdiff_t)max(u - 5 + c, 0) * Hc + (v - 5);
                const DwordQuad a = *reinterpret_cast<const DwordQuad *>(col);
                nw[c][0] = a.x, nw[c][1] = a.y, nw[c][2] = a.z, nw[c][3] = a.w;
                if (c < 5) {
                    const DwordQuad b = *reinterpret_cast<const DwordQuad *>(col + 4), e = *reinterpret_cast<const DwordQuad *>(col + 8);
                    nw[c][4] = b.x, nw[c][5] = b.y, nw[c][6] = b.z, nw[c][7] = b.w;
                    nw[c][8] = e.x, nw[c][9] = e.y, nw[c][10] = e.z, nw[c][11] = e.w;
                } else {
                    nw[c][4] = col[4];
                }
            }
            int sure = 0, maybe = 0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int du = c - 5, u2 = u + du;
                const bool col_ok = du >= -win && u2 >= 0;
#pragma unroll
                for (int r = 0; r < 11; r++) {
                    const int dv = r - 5;
                    if (du == 0 && dv >= 0) continue;
                    const int v2 = v + dv;
                    const bool ok = col_ok && dv >= -win && dv <= win && v2 >= 0 && v2 < Hc;
                    const int d2 = cs_value(nw[c][r]), s2 = cs_state(nw[c][r]) & 3;
                    const int cons = ok & (d2 >= 0) & (abs(dd - d2) <= thr);
                    sure += cons & (s2 == FST_KEEP);
                    maybe += cons & (s2 == FST_UNC);
                }
            }
            if (c_late + sure >= need)
                __hip_atomic_store(&stb[4 * (size_t)idx], (uint8_t)FST_KEEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (c_late + sure + maybe < need)
                __hip_atomic_store(&stb[4 * (size_t)idx], (uint8_t)FST_DROP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                open++;
        }
        __threadfence_block();
        left = __syncthreads_or(open);
    }
    if (!left) return;
    // what is still uncertain, in order
    int n_rest = 0;
    {
        const int per_q = (n_unc + RSV_THREADS - 1) / RSV_THREADS;
        const int q_lo = min(tid * per_q, n_unc), q_hi = min(q_lo + per_q, n_unc);
        int open = 0;
        for (int q = q_lo; q < q_hi; q++) open += (stb[4 * (size_t)list[q]] & 3) == FST_UNC ? 1 : 0;
        int p2 = block_exclusive_scan<RSV_THREADS>(open, s_wave, &n_rest);
        if (open)
            for (int q = q_lo; q < q_hi; q++)
                if ((stb[4 * (size_t)list[q]] & 3) == FST_UNC) rest[p2++] = list[q];
        __threadfence_block();
        __syncthreads();
    }
    // ... resolved in scan order by one wavefront: lanes = the earlier neighbours (win*(2win+1) + win <= 60 for win <= 5)
    if (tid < 64) {
        const int rowlen = 2 * win + 1, n_early = win * rowlen + win;
        for (int i = 0; i < n_rest; i++) {
            const int idx = (int)rest[i];
            const int u = idx / Hc, v = idx - u * Hc;
            const uint32_t w0 = cs[idx];  // (its own state is only written below)
            const int dd = cs_value(w0), c_late = cs_state(w0) >> 2;
            bool ok = false;
            if (lane < n_early) {
                const int u2 = lane < win * rowlen ? u - win + lane / rowlen : u;
                const int v2 = lane < win * rowlen ? v - win + lane % rowlen : v - win + (lane - win * rowlen);
                if (u2 >= 0 && v2 >= 0 && v2 < Hc) {
                    const uint32_t w2 = __hip_atomic_load(&cs[u2 * Hc + v2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int d2 = cs_value(w2);
                    ok = d2 >= 0 && abs(dd - d2) <= thr && (cs_state(w2) & 3) == FST_KEEP;
                }
            }
            const int c = (int)__popcll(__ballot(ok));
            if (lane == 0) __hip_atomic_store(&stb[4 * (size_t)idx], (uint8_t)((c_late + c >= need) ? FST_KEEP : FST_DROP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}

// ---- (3) / (4) redundancy passes (elas.cpp:178-233; max_dist 5, threshold 1).  Only points of one line interact, and the scan
// along a line is a recurrence over the kept-flags of the previous five points:
//     kept(p) = valid(p) && !( hi(p) && (S & m(p)) != 0 ),   S = kept-flags of p-1..p-5 (bit j-1 = point p-j)
// with hi(p) = "one of the five LATER points (still unmodified when p is visited) matches" and m(p) bit j-1 = "point p-j matches"
// (a match: valid and |difference| <= 1) - both from the pass's input, i.e. computable for every point independently.
// flag byte: bit 0 valid, bit 1 hi, bits 2-6 m.  `a` = the eleven values p-5 .. p+5 (out-of-line slots hold -1).
__device__ __forceinline__ uint32_t redundancy_flags(const int a[11]) {
    const int dd = a[5];
    if (dd < 0) return 0u;
    uint32_t hi = 0u, m = 0u;
#pragma unroll
    for (int j = 1; j <= 5; j++) {
        m |= (uint32_t)((a[5 - j] >= 0) & (abs(dd - a[5 - j]) <= 1)) << (j - 1);
        hi |= (uint32_t)((a[5 + j] >= 0) & (abs(dd - a[5 + j]) <= 1));
    }
    return 1u | (hi << 1) | (m << 2);
}

// Flag bytes of one line are contiguous ([line][LS] with LS = 4 (mod 16): lanes that walk neighbouring lines read different
// banks).  The walk reads 16 flags at a time - four independent dword reads, nothing in the loop is stored to the flags - and
// leaves the kept-flags as one 16-bit word per 16 positions in kb[line * KS + p / 16].
__host__ __device__ inline int filter_line_stride(int n) { return ((n + 15) & ~15) + 4; }
__host__ __device__ inline int filter_kept_stride(int n) { return ((n + 15) >> 4) | 1; }
__device__ __forceinline__ void redundancy_walk(const uint8_t *f, int n, uint16_t *kb) {
    const uint32_t *fw = reinterpret_cast<const uint32_t *>(f);
    uint32_t S = 0u;
    for (int p0 = 0; p0 < n; p0 += 16) {  // (positions beyond n: the line's padding holds zero flags)
        const uint32_t w0 = fw[(p0 >> 2)], w1 = fw[(p0 >> 2) + 1], w2 = fw[(p0 >> 2) + 2], w3 = fw[(p0 >> 2) + 3];
        uint32_t bits = 0u;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t w = j < 4 ? w0 : j < 8 ? w1 : j < 12 ? w2 : w3;
            const uint32_t x = (w >> (8 * (j & 3))) & 0xFFu;
            const uint32_t kept = (x & 1u) & ~(((x >> 1) & 1u) & (uint32_t)((S & (x >> 2)) != 0u));
            bits |= kept << j;
            S = ((S << 1) | kept) & 31u;
        }
        kb[p0 >> 4] = (uint16_t)bits;
    }
}

#define FRD_THREADS 256
// vertical: a strip of `SW` lattice columns (a contiguous index range).  LDS: values [SW][Hc + 10] int16, flags [SW][LS], kept
// words [SW][KS].  Input = the raw lattice minus the points the inconsistency pass dropped.
__global__ __launch_bounds__(FRD_THREADS) void k_filter_vertical(Dims d, int SW, const uint32_t *__restrict__ fcs, int16_t *__restrict__ latB) {
    extern __shared__ int16_t frd_lds[];
    const int pair = blockIdx.y, Wc = d.Wc, Hc = d.Hc, P = Hc + 10, LS = filter_line_stride(Hc), KS = filter_kept_stride(Hc);
    const int u0 = blockIdx.x * SW, nu = min(SW, Wc - u0);
    int16_t *val = frd_lds;
    uint16_t *kb = reinterpret_cast<uint16_t *>(val + (size_t)SW * P + ((SW * P) & 1));
    uint8_t *fl = reinterpret_cast<uint8_t *>(kb + (size_t)SW * KS + ((SW * KS) & 1));
    const size_t g0 = (size_t)pair * Wc * Hc + (size_t)u0 * Hc;
    const uint32_t *cs = fcs + (size_t)pair * filter_cs_stride(Wc * Hc) + FCS_PAD + (size_t)u0 * Hc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < nu; c += FRD_THREADS / 64) {
        for (int v = lane; v < Hc; v += 64) {
            const uint32_t w = cs[(size_t)c * Hc + v];
            val[c * P + 5 + v] = (cs_state(w) & 3) == FST_DROP ? (int16_t)-1 : (int16_t)cs_value(w);
        }
        if (lane < 10) val[c * P + (lane < 5 ? lane : Hc + lane)] = (int16_t)-1;
    }
    __syncthreads();
    for (int c = wave; c < nu; c += FRD_THREADS / 64)
        for (int v = lane; v < LS; v += 64) {  // (up to the end of the padded line: zero flags behind the last point)
            uint32_t f = 0u;
            if (v < Hc) {
                int a[11];
#pragma unroll
                for (int j = 0; j < 11; j++) a[j] = val[c * P + v + j];
                f = redundancy_flags(a);
            }
            fl[c * LS + v] = (uint8_t)f;
        }
    __syncthreads();
    if ((int)threadIdx.x < nu) redundancy_walk(fl + threadIdx.x * LS, Hc, kb + threadIdx.x * KS);
    __syncthreads();
    for (int c = wave; c < nu; c += FRD_THREADS / 64)
        for (int v = lane; v < Hc; v += 64) latB[g0 + (size_t)c * Hc + v] = ((kb[c * KS + (v >> 4)] >> (v & 15)) & 1) ? val[c * P + 5 + v] : (int16_t)-1;
}

// horizontal: a strip of `SH` lattice rows (SH a power of two <= 64).  LDS: values [Wc + 10][SH] int16, flags [SH][LS], kept
// words [SH][KS].
__global__ __launch_bounds__(FRD_THREADS) void k_filter_horizontal(Dims d, int SH, const int16_t *__restrict__ latB, int16_t *__restrict__ latC, int32_t *bcnt, int nb2) {
    extern __shared__ int16_t frd_lds[];
    const int pair = blockIdx.y, Wc = d.Wc, Hc = d.Hc, LS = filter_line_stride(Wc), KS = filter_kept_stride(Wc);
    const int v0 = blockIdx.x * SH, nv = min(SH, Hc - v0);
    int16_t *val = frd_lds;
    uint16_t *kb = reinterpret_cast<uint16_t *>(val + (size_t)(Wc + 10) * SH + (((Wc + 10) * SH) & 1));
    uint8_t *fl = reinterpret_cast<uint8_t *>(kb + (size_t)SH * KS + ((SH * KS) & 1));
    int *hist = reinterpret_cast<int *>(fl + (((size_t)SH * LS + 3) & ~(size_t)3));  // [nb2] points this strip adds to each collect block
    for (int i = threadIdx.x; i < nb2; i += FRD_THREADS) hist[i] = 0;
    const size_t g0 = (size_t)pair * Wc * Hc + v0;
    const int r = threadIdx.x & (SH - 1), ug = threadIdx.x / SH, ustep = FRD_THREADS / SH;
    for (int u = ug - 5; u < Wc + 5; u += ustep) val[(u + 5) * SH + r] = (u >= 0 && u < Wc && r < nv) ? latB[g0 + (size_t)u * Hc + r] : (int16_t)-1;
    __syncthreads();
    for (int u = ug; u < LS; u += ustep) {
        uint32_t f = 0u;
        if (u < Wc) {
            int a[11];
#pragma unroll
            for (int j = 0; j < 11; j++) a[j] = val[(u + j) * SH + r];
            f = redundancy_flags(a);
        }
        fl[r * LS + u] = (uint8_t)f;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) redundancy_walk(fl + threadIdx.x * LS, Wc, kb + threadIdx.x * KS);
    __syncthreads();
    if (r < nv)
        for (int u = ug; u < Wc; u += ustep) {
            const bool kept = (kb[r * KS + (u >> 4)] >> (u & 15)) & 1;
            latC[g0 + (size_t)u * Hc + r] = kept ? val[(u + 5) * SH + r] : (int16_t)-1;
            // the collection (elas.cpp:424-428) leaves out lattice row 0 and column 0; its blocks are FLT_THREADS consecutive indices
            if (kept && u >= 1 && v0 + r >= 1) atomicAdd(&hist[(u * Hc + v0 + r) / FLT_THREADS], 1);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < nb2; i += FRD_THREADS)
        if (hist[i]) atomicAdd(&bcnt[(size_t)pair * nb2 + i], hist[i]);
}

// ---- (5) collection in scan order (elas.cpp:424-428; lattice row / column 0 excluded) and corner points (elas.cpp:235-264), as
// blocks of FLT_THREADS consecutive lattice indices - kernel boundaries are the only synchronisation.  The horizontal pass
// leaves the number of points each block will emit (bcnt, cleared by the resolve step).  k_filter_collect: a block's first list position is the sum
// of its predecessors' counts (<= 1 300 values even for the largest lattice); it writes its points and offers the nearest
// one per image corner as a key (distance^2, list position, disparity) - "first minimum in list order" (:246-253) is the
// minimum of (distance, position); the pair's block 0 of a third, tiny launch (k_filter_corners) reduces the keys and appends
// the six corner points.
__device__ __forceinline__ bool collect_pred(const Dims &d, const int16_t *__restrict__ C, int i, int lat, int &u, int &v, int &dv) {
    dv = -1;
    u = v = 0;
    if (i < lat) {
        u = i / d.Hc;
        v = i - u * d.Hc;
        dv = C[i];
    }
    return dv >= 0 && u >= 1 && v >= 1;
}

__global__ __launch_bounds__(FLT_THREADS) void k_filter_collect(KParams k, const int16_t *__restrict__ latC, const int32_t *__restrict__ bcnt, unsigned long long *__restrict__ bkey,
                                                                int32_t *__restrict__ fsup) {
    const Dims &d = k.d;
    const int pair = blockIdx.y, b = blockIdx.x, nb2 = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lat = d.Wc * d.Hc;
    __shared__ int s_wave[FLT_THREADS / 64];
    __shared__ int s_excl;
    __shared__ unsigned long long s_key[4][FLT_THREADS / 64];
    int u, v, dv;
    const bool pred = collect_pred(d, latC + (size_t)pair * lat, b * FLT_THREADS + tid, lat, u, v, dv);
    const unsigned long long m = __ballot(pred);
    if (lane == 0) s_wave[wave] = (int)__popcll(m);
    if (wave == 0) {  // this block's first list position: the sum of its predecessors' counts
        int acc = 0;
        for (int j = lane; j < b; j += 64) acc += bcnt[(size_t)pair * nb2 + j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) s_excl = acc;
    }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < FLT_THREADS / 64; w++) base += w < wave ? s_wave[w] : 0;
    const int q = s_excl + base + (int)__popcll(m & ((1ull << lane) - 1ull));
    int32_t *out = fsup + (size_t)pair * d.max_pts * 3;
    if (pred) {
        out[3 * q] = u * d.step;
        out[3 * q + 1] = v * d.step;
        out[3 * q + 2] = dv;
    }
    if (!k.add_corners) return;
    const int bu[4] = {0, 0, d.W - 1, d.W - 1}, bv[4] = {0, d.H - 1, 0, d.H - 1};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        unsigned long long key = ~0ull;
        if (pred) {
            const int du = bu[c] - u * d.step, dw = bv[c] - v * d.step;
            const int dist = du * du + dw * dw;
            // (the reference's search starts from best_dist = 10000000, :245: points that far away are never taken)
            if (dist < 10000000) key = ((unsigned long long)dist << 34) | ((unsigned long long)q << 11) | (unsigned long long)dv;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off, 64);
            key = o < key ? o : key;
        }
        if (lane == 0) s_key[c][wave] = key;
    }
    __syncthreads();
    if (tid < 4) {
        unsigned long long key = ~0ull;
#pragma unroll
        for (int w = 0; w < FLT_THREADS / 64; w++) key = s_key[tid][w] < key ? s_key[tid][w] : key;
        bkey[((size_t)pair * nb2 + b) * 4 + tid] = key;
    }
}

// one wavefront per pair: point count, corner points (elas.cpp:235-264), fnsup
__global__ __launch_bounds__(64) void k_filter_corners(KParams k, const int32_t *__restrict__ bcnt, const unsigned long long *__restrict__ bkey, int nb2, int32_t *__restrict__ fsup,
                                                       int32_t *__restrict__ fnsup) {
    const Dims &d = k.d;
    const int pair = blockIdx.x, lane = threadIdx.x;
    int n_main = 0;
    unsigned long long key[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    for (int j = lane; j < nb2; j += 64) {
        n_main += bcnt[(size_t)pair * nb2 + j];
        if (k.add_corners)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const unsigned long long o = bkey[((size_t)pair * nb2 + j) * 4 + c];
                key[c] = o < key[c] ? o : key[c];
            }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        n_main += __shfl_xor(n_main, off, 64);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const unsigned long long o = __shfl_xor(key[c], off, 64);
            key[c] = o < key[c] ? o : key[c];
        }
    }
    if (lane != 0) return;
    int n_total = n_main;
    if (k.add_corners) {
        int32_t *out = fsup + (size_t)pair * d.max_pts * 3;
        const int bu[4] = {0, 0, d.W - 1, d.W - 1}, bv[4] = {0, d.H - 1, 0, d.H - 1};
        int cd[4];
        for (int c = 0; c < 4; c++) cd[c] = key[c] == ~0ull ? 0 : (int)(key[c] & 0x7FFull);
        int qq = n_main;
        for (int c = 0; c < 4; c++) {
            out[3 * qq] = bu[c];
            out[3 * qq + 1] = bv[c];
            out[3 * qq + 2] = cd[c];
            qq++;
        }
        for (int c = 2; c < 4; c++) {  // the two right-image corners (:258-259)
            out[3 * qq] = bu[c] + cd[c];
            out[3 * qq + 1] = bv[c];
            out[3 * qq + 2] = cd[c];
            qq++;
        }
        n_total = n_main + 6;
    }
    fnsup[pair] = n_total;
}

// LDS bytes of the redundancy kernels for strips of `s` lines of `n` points
static size_t filter_strip_lds(int s, int n) {
    return (size_t)s * (n + 10) * 2 + 2 + ((size_t)s * filter_kept_stride(n) + 1) * 2 + (size_t)s * filter_line_stride(n);
}
// strip width: 8 lines (many workgroups, short flag phases: 11.5 + 20 us per 32-pair KITTI launch for the two passes against 14 + 22
// with 16 lines and 18 + 33 with 32), fewer when a line is so long that the tile would exceed 96 KB
static int filter_strip(int line_len) {
    int s = 8;
    while (s > 1 && filter_strip_lds(s, line_len) > 96 * 1024) s >>= 1;
    return s;
}

size_t support_filter_ws_bytes(const KParams &k, int cap) {  // per slot: segment + ordered list of uncertain points, two lattice copies, state bytes, block counts and keys
    const size_t lat = (size_t)k.d.Wc * k.d.Hc, nb = (lat + FCL_THREADS - 1) / FCL_THREADS, nb2 = (size_t)filter_collect_blocks((int)lat);
    return (size_t)cap * (lat * (4 + 4 + 2 + 2) + filter_cs_stride((int)lat) * 4 + nb * 4 + nb2 * (4 * 8 + 4) + 16) + 256;
}

void launch_support_filter(const KParams &k, int win, int thr, int need, const SlotDev &s, int n, hipStream_t st) {
    const size_t lat = (size_t)k.d.Wc * k.d.Hc, cap = (size_t)s.cap;
    const int nb = (int)((lat + FCL_THREADS - 1) / FCL_THREADS), nb2 = filter_collect_blocks((int)lat);
    uint8_t *base = static_cast<uint8_t *>(s.flt_ws);
    unsigned long long *bkey = reinterpret_cast<unsigned long long *>(base);
    uint32_t *useg = reinterpret_cast<uint32_t *>(bkey + cap * (size_t)nb2 * 4), *ulist = useg + cap * lat;
    int32_t *ucnt = reinterpret_cast<int32_t *>(ulist + cap * lat), *bcnt = ucnt + cap * (size_t)nb;
    uint32_t *fcs = reinterpret_cast<uint32_t *>(bcnt + cap * (size_t)nb2);  // value | state << 16 per point, padded per pair
    int16_t *latB = reinterpret_cast<int16_t *>(fcs + cap * filter_cs_stride((int)lat)), *latC = latB + cap * lat;
    const size_t cl_lds = sizeof(int16_t) * (size_t)((FCL_THREADS + k.d.Hc - 1) / k.d.Hc + 12) * (size_t)(k.d.Hc + 10);  // columns a block and its windows touch, padded rows
    const int SW = filter_strip(k.d.Hc), SH = filter_strip(k.d.Wc);
    const size_t v_lds = filter_strip_lds(SW, k.d.Hc), h_lds = filter_strip_lds(SH, k.d.Wc) + 8 + sizeof(int) * (size_t)nb2;
    static std::atomic<size_t> granted[64], granted_v[64], granted_h[64];
    ensure_dynamic_lds(k_filter_classify, cl_lds, granted, "support_filter");
    ensure_dynamic_lds(k_filter_vertical, v_lds, granted_v, "support_filter (vertical)");
    ensure_dynamic_lds(k_filter_horizontal, h_lds, granted_h, "support_filter (horizontal)");
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_classify, dim3((unsigned)nb, n), dim3(FCL_THREADS), cl_lds, st, k.d, win, thr, need, s.dcan, fcs, useg, ucnt);
    if (nb > RSV_MAX_BLOCKS) throw std::runtime_error("support_filter: lattice too large for the resolve step's block table");
    // (urest: the remaining points after the rounds; they overwrite nothing the rounds' list still needs - a buffer of its own: latC is free until the horizontal pass)
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_resolve, dim3(n), dim3(RSV_THREADS), sizeof(int) * ((size_t)nb + 1), st, k.d, win, thr, need, fcs, useg, ucnt, nb, ulist,
              reinterpret_cast<uint32_t *>(latB), bcnt, nb2);
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_vertical, dim3((k.d.Wc + SW - 1) / SW, n), dim3(FRD_THREADS), v_lds, st, k.d, SW, fcs, latB);
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_horizontal, dim3((k.d.Hc + SH - 1) / SH, n), dim3(FRD_THREADS), h_lds, st, k.d, SH, latB, latC, bcnt, nb2);
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_collect, dim3(nb2, n), dim3(FLT_THREADS), 0, st, k, latC, bcnt, bkey, s.fsup);
    SV_LAUNCH(K_SUPPORT_FILTER, k_filter_corners, dim3(n), dim3(64), 0, st, k, bcnt, bkey, nb2, s.fsup, s.fnsup);
}

// ------------------------------------------------------------------------------------------------------------
// K3  candidate grid as per-cell disparity bit masks
//     reference: elas.cpp:577-653 (createGrid); the 3x3 dilation runs over the FLAT cell array (:613-628)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_mark(KParams k, const int32_t *__restrict__ blob, uint32_t *__restrict__ gA) {
    const Dims &d = k.d;
    const int pair = blockIdx.z, side = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = blob[pair * META_WORDS];
    if (n < 3 || i >= n) return;
    const int32_t *p = blob + blob[pair * META_WORDS + 1] + (size_t)i * 3;
    const int x_curr = p[0], y_curr = p[1], d_curr = p[2];
    const int d_min = max(d_curr - 1, 0), d_max = min(d_curr + 1, d.disp_max);
    int x;
    if (side == 0)
        x = (int)floorf((float)(x_curr / d.grid_size));  // integer division first (:599)
    else
        x = (int)floorf((float)(x_curr - d_curr) / (float)d.grid_size);
    const int y = (int)floorf((float)y_curr / (float)d.grid_size);
    if (x < 0 || x >= d.gw || y < 0 || y >= d.gh) return;
    uint32_t *cell = gA + (((size_t)(pair * 2 + side) * d.ncell) + (size_t)y * d.gw + x) * d.MW;
    for (int dd = d_min; dd <= d_max; dd++) atomicOr(&cell[dd >> 5], 1u << (dd & 31));
}

__global__ __launch_bounds__(256) void k_grid_dilate(KParams k, const uint32_t *__restrict__ gA, uint32_t *__restrict__ gB) {
    const Dims &d = k.d;
    const int ps = blockIdx.y;  // pair*2 + side
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.ncell * d.MW) return;
    const int c = i / d.MW, w = i - c * d.MW;
    const uint32_t *in = gA + (size_t)ps * d.ncell * d.MW;
    uint32_t r = 0;
    if (c >= d.gw + 1 && c <= d.ncell - d.gw - 2) {
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) r |= in[(size_t)(c + dy * d.gw + dx) * d.MW + w];
    }
    gB[(size_t)ps * d.ncell * d.MW + i] = r;
}

size_t grid_masks_words(const KParams &k, int cap) { return (size_t)cap * 2 * k.d.ncell * k.d.MW; }
size_t raster_tiles(const KParams &k) { return (size_t)((k.d.W + 63) / 64) * ((k.d.H + 31) / 32); }  // RT_W x RT_H tiles
size_t grid_clear_bytes(const KParams &k, int cap) { return sizeof(uint32_t) * (grid_masks_words(k, cap) + (size_t)cap * 2 * raster_tiles(k)); }

// max_points: the largest support-point count of the chunk as the host knows it (it wrote the blob); the grids of the per-point
// and per-triangle kernels are sized by it instead of by the capacity (a KITTI lattice holds 18 681 points, a pair has ~2 100)
void launch_grid(const KParams &k, const SlotDev &s, int n, int max_points, hipStream_t st) {
    // one clear for the cell masks and the raster tile counters of the whole slot (they are one allocation: gmaskA, tile_cnt)
    (void)hipMemsetAsync(s.gmaskA, 0, grid_clear_bytes(k, s.cap), st);
    const int np = std::max(1, std::min(max_points, k.d.max_pts));
    SV_LAUNCH(K_GRID_MARK, k_grid_mark, dim3((np + 255) / 256, 2, n), dim3(256), 0, st, k, s.blob, s.gmaskA);
    SV_LAUNCH(K_GRID_DILATE, k_grid_dilate, dim3((k.d.ncell * k.d.MW + 255) / 256, n * 2), dim3(256), 0, st, k, s.gmaskA, s.gmaskB);
}

// ------------------------------------------------------------------------------------------------------------
// K4  per-triangle plane fit + scan conversion
//     reference: elas.cpp:503-575 (computeDisparityPlanes) -> common_includes/elas/matrix.cpp:418-510 (Gauss-Jordan,
//                full pivoting, double); elas.cpp:839-941 (corner sort, edge lines, scan conversion order)
//     One wavefront per triangle: every lane solves the two 3x3 systems (uniform work), then lanes stride over the
//     triangle's columns and mark covered pixels with atomicMax(triangle index) so that, as in the reference's
//     sequential loop, the LAST triangle covering a pixel decides it.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool solve3(double A[3][3], double B[3]) {
    int ipiv[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double big = 0.0;
        int irow = 0, icol = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (ipiv[j] != 1)
#pragma unroll
                for (int kk = 0; kk < 3; kk++)
                    if (ipiv[kk] == 0)
                        if (fabs(A[j][kk]) >= big) {
                            big = fabs(A[j][kk]);
                            irow = j;
                            icol = kk;
                        }
        // ++ipiv[icol], row swap and elimination with runtime row indices, written with selects so the 3x3 stays in registers
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j == icol) ipiv[j]++;
        if (irow != icol) {
#pragma unroll
            for (int l = 0; l < 3; l++) {
                double x = 0, y = 0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (j == irow) x = A[j][l];
                    if (j == icol) y = A[j][l];
                }
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    if (j == irow) A[j][l] = y;
                    if (j == icol) A[j][l] = x;
                }
            }
            double x = 0, y = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j == irow) x = B[j];
                if (j == icol) y = B[j];
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j == irow) B[j] = y;
                if (j == icol) B[j] = x;
            }
        }
        double piv = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j == icol) piv = A[j][j];
        if (fabs(piv) < 1e-20) return false;
        const double pivinv = 1.0 / piv;
        double prow[3] = {0, 0, 0}, pb = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j == icol) {
                A[j][j] = 1.0;
#pragma unroll
                for (int l = 0; l < 3; l++) {
                    A[j][l] *= pivinv;
                    prow[l] = A[j][l];
                }
                B[j] *= pivinv;
                pb = B[j];
            }
#pragma unroll
        for (int ll = 0; ll < 3; ll++)
            if (ll != icol) {
                double dum = 0;
#pragma unroll
                for (int l = 0; l < 3; l++)
                    if (l == icol) {
                        dum = A[ll][l];
                        A[ll][l] = 0.0;
                    }
#pragma unroll
                for (int l = 0; l < 3; l++) A[ll][l] -= prow[l] * dum;
                B[ll] -= pb * dum;
            }
    }
    return true;
}

// per-triangle record handed from k_planes to k_raster: column ranges and the three edge lines (elas.cpp:887-906)
struct RasterRec {
    int a_u, b_u, c_u;          // (int32_t)A_u, B_u, C_u
    float ac_a, ac_b, ab_a, ab_b, bc_a, bc_b;
};

// Scan conversion goes through image tiles held in LDS: k_planes bins every triangle into the tiles its bounding box
// touches, k_raster_tiles resolves "the last triangle in list order that covers a pixel decides it" with LDS atomicMax and
// writes each tile with coalesced stores.  A tile list that overflows switches its (pair, side) to the global-atomic path.
#define RT_W 64
#define RT_H 32
#define RT_CAP 512

__global__ __launch_bounds__(256) void k_planes(KParams k, const int32_t *__restrict__ blob, float4 *__restrict__ trirec, float *__restrict__ planes,
                                                RasterRec *__restrict__ rrec, int32_t *__restrict__ tile_cnt, int32_t *__restrict__ tile_list) {
    const Dims &d = k.d;
    const int pair = blockIdx.z, side = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int32_t *meta = blob + pair * META_WORDS;
    if (meta[0] < 3) return;  // (the whole workgroup)
    // Tile binning goes through the workgroup: counted per tile in LDS, one global atomic per (workgroup, tile) reserves the
    // list slots, ranks inside the reservation again from LDS.  A workgroup's 256 consecutive triangles fall into a handful of
    // tiles; with one global atomic per (triangle, tile) the ~200 increments per tile counter queued up at the L2 and the binning
    // took twice as long as the two f64 solves (43 of the kernel's 64 us per 32-pair launch).
    extern __shared__ int32_t pl_lds[];  // [2][ntile]: counts / ranks, reserved bases
    const int ntx = (d.W + RT_W - 1) / RT_W, nty = (d.H + RT_H - 1) / RT_H, ntile = ntx * nty;
    int32_t *s_cnt = pl_lds, *s_base = pl_lds + ntile;
    for (int i = threadIdx.x; i < ntile; i += 256) s_cnt[i] = 0;
    __syncthreads();
    const bool active = t < meta[2 + 2 * side];
    int ub = 0, ue = 0, vb = 0, ve = 0;  // the triangle's tile footprint (empty for idle threads)
    if (active) {
    const size_t tbase = ((size_t)(pair * 2 + side) * d.max_tri + t);
    const int32_t *tc = blob + meta[3 + 2 * side] + (size_t)t * 3;
    const int32_t *support = blob + meta[1];
    int pu[3], pv[3], pd[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int32_t *p = support + (size_t)tc[c] * 3;
        pu[c] = p[0];
        pv[c] = p[1];
        pd[c] = p[2];
    }
    float pl[6];
#pragma unroll
    for (int sd = 0; sd < 2; sd++) {  // sd 0: left-image plane (t1*), sd 1: right-image plane (t2*)
        double A[3][3], B[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            A[r][0] = sd == 0 ? (double)pu[r] : (double)(pu[r] - pd[r]);
            A[r][1] = (double)pv[r];
            A[r][2] = 1.0;
            B[r] = (double)pd[r];
        }
        const bool ok = solve3(A, B);
        pl[3 * sd + 0] = ok ? (float)B[0] : 0.f;
        pl[3 * sd + 1] = ok ? (float)B[1] : 0.f;
        pl[3 * sd + 2] = ok ? (float)B[2] : 0.f;
    }
    const float plane_a = side == 0 ? pl[0] : pl[3], plane_b = side == 0 ? pl[1] : pl[4], plane_c = side == 0 ? pl[2] : pl[5];
    const float plane_d = side == 0 ? pl[3] : pl[0];
    const bool valid = (double)fabsf(plane_a) < 0.7 && (double)fabsf(plane_d) < 0.7;  // elas.cpp:910
    trirec[tbase] = make_float4(plane_a, plane_b, plane_c, valid ? 1.0f : 0.0f);
#pragma unroll
    for (int j = 0; j < 6; j++) planes[tbase * 6 + j] = pl[j];
    // corner sort wrt u, ascending (elas.cpp:859-884)
    float tu[3], tv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        tu[c] = side == 0 ? (float)pu[c] : (float)(pu[c] - pd[c]);
        tv[c] = (float)pv[c];
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int kk = 0; kk < j; kk++)
            if (tu[kk] > tu[j]) {
                float x = tu[j];
                tu[j] = tu[kk];
                tu[kk] = x;
                x = tv[j];
                tv[j] = tv[kk];
                tv[kk] = x;
            }
    const float A_u = tu[0], A_v = tv[0], B_u = tu[1], B_v = tv[1], C_u = tu[2], C_v = tv[2];
    float AB_a = 0, AC_a = 0, BC_a = 0;  // :894-906
    if ((int)A_u != (int)B_u) AB_a = (A_v - B_v) / (A_u - B_u);
    if ((int)A_u != (int)C_u) AC_a = (A_v - C_v) / (A_u - C_u);
    if ((int)B_u != (int)C_u) BC_a = (B_v - C_v) / (B_u - C_u);
    RasterRec r;
    r.a_u = (int)A_u;
    r.b_u = (int)B_u;
    r.c_u = (int)C_u;
    r.ac_a = AC_a;
    r.ac_b = A_v - AC_a * A_u;
    r.ab_a = AB_a;
    r.ab_b = A_v - AB_a * A_u;
    r.bc_a = BC_a;
    r.bc_b = B_v - BC_a * B_u;
    rrec[tbase] = r;
    // bin into tiles: columns [max(A_u,0), min(C_u,W)), rows between the corners' v (+-1 for the float->int truncations)
    ub = max(r.a_u, 0), ue = min(r.c_u, d.W);
    const float vlo = fminf(fminf(A_v, B_v), C_v), vhi = fmaxf(fmaxf(A_v, B_v), C_v);
    vb = max((int)vlo - 1, 0), ve = min((int)vhi + 2, d.H);  // rows [vb, ve)
    }
    const bool bins = ub < ue && vb < ve;
    const int tx_lo = ub / RT_W, tx_hi = bins ? (ue - 1) / RT_W : -1, ty_lo = vb / RT_H, ty_hi = bins ? (ve - 1) / RT_H : -1;
    for (int ty = ty_lo; ty <= ty_hi; ty++)
        for (int tx = tx_lo; tx <= tx_hi; tx++) atomicAdd(&s_cnt[ty * ntx + tx], 1);
    __syncthreads();
    const size_t tb = (size_t)(pair * 2 + side) * ntile;
    for (int i = threadIdx.x; i < ntile; i += 256) {
        const int c = s_cnt[i];
        if (c > 0) s_base[i] = atomicAdd(&tile_cnt[tb + i], c);
        s_cnt[i] = 0;
    }
    __syncthreads();
    for (int ty = ty_lo; ty <= ty_hi; ty++)
        for (int tx = tx_lo; tx <= tx_hi; tx++) {
            const int tile = ty * ntx + tx;
            const int slot = s_base[tile] + atomicAdd(&s_cnt[tile], 1);
            if (slot < k.rt_cap) tile_list[(tb + tile) * RT_CAP + slot] = t;  // (a tile whose count passes the cap walks all triangles)
        }
}

// Scan conversion (elas.cpp:912-940): atomicMax(triangle index) on the tile reproduces "the last triangle in list order that
// covers a pixel decides it".

// elas.cpp:912-940 for the part of every binned triangle that falls into this workgroup's tile
__global__ __launch_bounds__(256) void k_raster_tiles(KParams k, const int32_t *__restrict__ blob, const RasterRec *__restrict__ rrec, const int32_t *__restrict__ tile_cnt,
                                                      const int32_t *__restrict__ tile_list, int32_t *__restrict__ tri_id) {
    const Dims &d = k.d;
    const int pair = blockIdx.z, side = blockIdx.y;
    if (blob[pair * META_WORDS] < 3) return;
    const int ntx = (d.W + RT_W - 1) / RT_W;
    const int tx0 = (blockIdx.x % ntx) * RT_W, ty0 = (blockIdx.x / ntx) * RT_H;
    __shared__ int32_t tile[RT_H][RT_W];
    for (int i = threadIdx.x; i < RT_H * RT_W; i += 256) (&tile[0][0])[i] = -1;
    __syncthreads();
    const size_t gt = (size_t)(pair * 2 + side) * gridDim.x + blockIdx.x;
    // A tile whose list overflowed (more than rt_cap triangles touch it: pathological triangulations) walks ALL triangles of its
    // side instead - those outside the tile clip to nothing.  Decided per tile: no second kernel, no flag.
    const int listed = tile_cnt[gt];
    const bool all = listed > k.rt_cap;
    const int cnt = all ? blob[pair * META_WORDS + 2 + 2 * side] : listed;
    // 8 lanes per triangle (a lattice triangle is ~5 columns wide), one lane per column of [A_u, C_u): the edge below / above
    // the long edge AC is AB left of B_u and BC from B_u on (elas.cpp:913-925 and :928-940 are the two halves of this range)
    const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int x_end = min(tx0 + RT_W, d.W), y_end = min(ty0 + RT_H, d.H);
    // whole 8-lane groups walk the list together, one trip ahead with the loads (list entry, then its record: two dependent
    // memory accesses that would otherwise sit in front of every trip)
    const RasterRec *recs = rrec + (size_t)(pair * 2 + side) * d.max_tri;
    const int32_t *list = tile_list + gt * RT_CAP;
    int t_next = grp < cnt ? (all ? grp : list[grp]) : 0;
    RasterRec r_next = recs[t_next];
    for (int i = grp; i < cnt; i += 32) {
        const int t = t_next;
        const RasterRec r = r_next;
        if (i + 32 < cnt) {
            t_next = all ? i + 32 : list[i + 32];
            r_next = recs[t_next];
        }
        const int u_begin = max(max(r.a_u, 0), tx0), u_end = min(min(r.c_u, d.W), x_end);
        for (int base = u_begin; base < u_end; base += 8) {
            const int u = base + sub;
            int lo = 0x7FFFFFFF, hi = -0x7FFFFFFF;
            if (u < u_end) {
                const bool second = u >= r.b_u;
                const float e_a = second ? r.bc_a : r.ab_a, e_b = second ? r.bc_b : r.ab_b;
                const int v_1 = (int)(r.ac_a * (float)u + r.ac_b), v_2 = (int)(e_a * (float)u + e_b);
                lo = max(max(min(v_1, v_2), 0), ty0);
                hi = min(min(max(v_1, v_2), d.H), y_end);
                if (lo >= hi) {
                    lo = 0x7FFFFFFF;
                    hi = -0x7FFFFFFF;
                }
            }
            // every lane walks its own column piece [lo, hi) (nothing is shared between the lanes of a group here: a loop over the
            // group's common row range with a per-lane test cost two cross-lane reductions and two compares per row more)
            int32_t *p = lo < hi ? &tile[lo - ty0][u - tx0] : nullptr;
            for (int v = lo; v < hi; v++, p += RT_W) atomicMax(p, t);
        }
    }
    __syncthreads();
    int32_t *ids = tri_id + (size_t)(pair * 2 + side) * d.N;
    for (int i = threadIdx.x; i < RT_H * RT_W; i += 256) {
        const int r = i / RT_W, c = i - r * RT_W;
        if (ty0 + r < d.H && tx0 + c < d.W) map_st(ids, (uint32_t)((ty0 + r) * d.W + tx0 + c), tile[r][c]);
    }
}

void launch_triangles(const KParams &k, const SlotDev &s, int n, int max_points, hipStream_t st) {
    const int ntile = ((k.d.W + RT_W - 1) / RT_W) * ((k.d.H + RT_H - 1) / RT_H);
    // (the tile counters were cleared together with the cell masks: launch_grid)
    const int nt = std::max(1, std::min(2 * max_points, k.d.max_tri));  // a triangulation of p points has < 2p triangles
    const size_t pl_lds = sizeof(int32_t) * 2 * (size_t)ntile;
    static std::atomic<size_t> pl_granted[64];
    ensure_dynamic_lds(k_planes, pl_lds, pl_granted, "plane_fit");
    SV_LAUNCH(K_PLANES, k_planes, dim3((nt + 255) / 256, 2, n), dim3(256), pl_lds, st, k, s.blob, s.trirec, s.planes, (RasterRec *)s.rrec, s.tile_cnt, s.tile_list);
    SV_LAUNCH(K_TRIANGLES, k_raster_tiles, dim3(ntile, 2, n), dim3(256), 0, st, k, s.blob, (const RasterRec *)s.rrec, s.tile_cnt, s.tile_list, s.tri_id);
}

// ------------------------------------------------------------------------------------------------------------
// K5  dense matching (MAP over ~17 candidate disparities per pixel), both sides in one launch
//     reference: elas.cpp:688-801 (findMatch) + :655-686 (updatePosteriorMinimum), pixel set from :912-940
//     One lane per pixel; candidates = cell's grid mask outside the plane band (ascending), then the band
//     (ascending, + prior); strict '<' keeps the first minimum, exactly like the sequential reference.
// ------------------------------------------------------------------------------------------------------------
// One workgroup = DENSE_TW columns of one row, BOTH sides: left pixel u needs right columns [u-disp_max, u], right pixel u
// needs left columns [u, u+disp_max]; staging left[x0, x0+TW+disp_max) and right[x0-disp_max, x0+TW) once serves both
// passes and also holds every pixel's own descriptor, so each descriptor row segment is fetched ~(TW+disp_max)/TW times
// instead of ~2.5 times with one pass per launch.
#define DENSE_TW 512
#define DENSE_MASK_WORDS 8  // mask words per cell kept in LDS / registers (disp_max <= 255); further words are read from memory

// All 2R+1 band candidates of a pixel (slots pb[0..2R], see dense_pixel): relative keys (energy + prior) << 16 | o'
template <int R>
__device__ __forceinline__ int dense_band_full(const KParams &k, int side, const uint4 own, const uint4 *pb, uint32_t vmask) {
    uint4 c[2 * R + 1];
#pragma unroll
    for (int s = 0; s <= 2 * R; s++) c[s] = pb[s];
    int key[2 * R + 1];
#pragma unroll
    for (int s = 0; s <= 2 * R; s++) {
        const int o = side ? s : 2 * R - s, ao = s < R ? R - s : s - R;
        key[s] = sad16_key(own, c[s], (int)(vmask & (((uint32_t)k.prior[ao] << 16) | (uint32_t)o)));
    }
    int bb = key[0];
#pragma unroll
    for (int s = 1; s + 1 <= 2 * R; s += 2) bb = min(bb, min(key[s], key[s + 1]));  // v_min3
    return bb;
}

// MWT / RT: mask words per cell and plane radius as compile-time constants (0 = read them from the parameters).  The kernel
// has many wave-uniform decisions on them (which mask words exist, which band slots exist); as runtime values the compiler keeps
// ~60 scalar conditions alive per workgroup, spills them to VGPR lanes and reloads them in the pixel loops.
template <bool COUNT, int MWT, int RT, bool TEX>
__device__ __forceinline__ int dense_pixel(const KParams &k, int side, int u, int v, const uint4 own, const uint4 *pu, const float4 rec, const uint32_t *mw,
                                             const uint32_t *cell, int &ncand, int (&npath)[5], const uint4 *lds_first, const uint4 *lds_last) {
    const Dims &d = k.d;
    const int MW = MWT ? MWT : d.MW, plane_radius = RT ? RT : k.plane_radius;
    // elas.cpp:732-736 (the map keeps its -10).  TEX = false: match_texture <= 0 (MIDDLEBURY and the driver's preset), the sum of
    // absolute values can never be below it - the test and its four SADs are not compiled in (-4 % of the kernel)
    if (TEX && (int)texture16(own) < k.match_texture) return -10;
    const int d_plane = (int)(rec.x * (float)u + rec.y * (float)v + rec.z);      // :739, ((a*u)+(b*v))+c without contraction
    const int d_plane_min = max(d_plane - plane_radius, 0);
    const int d_plane_max = min(d_plane + plane_radius, d.D - 1);
    const bool valid = rec.w != 0.0f;
    // disparities whose warped column stays inside [2, W-3] (:763, :770 / :782, :789), as a range instead of a per-candidate test
    const int a_lo = side ? 0 : max(u - (d.W - 3), 0), a_hi = side ? min(d.W - 3 - u, d.D - 1) : min(u - 2, d.D - 1);
    const int sgn = side ? 1 : -1;  // candidate d lives at pu[-d] (left pixel) or pu[+d] (right pixel)
    // The reference keeps the FIRST minimum in evaluation order (strict <): grid candidates in ascending d, then the band in
    // ascending d.  Here every candidate yields the signed key  energy << 16 | band << 15 | d  (sad16_key) and the smallest key
    // wins: the same candidate, with one v_min_i32 per candidate instead of compare + two selects.
    constexpr int KEY_NONE = 0x7FFF0000;  // energy 32767: above every real energy (<= 4080) and the reference's initial 10000
    int best = KEY_NONE;
    const int b_lo = max(d_plane_min, a_lo), b_hi = min(d_plane_max, a_hi);
    // away from the image's left / right border every lane of the wavefront may use the whole range (mask bits above disp_max
    // are never set): the column-range clipping of the masks is skipped then
    const bool clip = __builtin_amdgcn_ballot_w64(a_lo != 0 || a_hi != d.D - 1) != 0;
    // the band (at most 2 * plane_radius + 1 <= 31 bits) as a 64-bit mask starting in mask word band_word
    // (the ones - at most 2 * 15 + 1 = 31 of them, none for an empty band - in 32 bits, then ONE 64-bit shift puts them in place)
    const int band_bits = max(d_plane_max - d_plane_min + 1, 0), band_word = d_plane_min >> 5;
    const uint64_t band = (uint64_t)((1u << band_bits) - 1u) << (d_plane_min & 31);
    const uint32_t band_lo = (uint32_t)band, band_hi = (uint32_t)(band >> 32);
    uint32_t mc[DENSE_MASK_WORDS];
#pragma unroll
    for (int w = 0; w < DENSE_MASK_WORDS; w++) mc[w] = mw[w];
    if (clip) {  // image border only: keep [a_lo, a_hi] (one wave-uniform branch for all words)
#pragma unroll
        for (int w = 0; w < DENSE_MASK_WORDS; w++) {
            if (w >= MW) break;
            const int lo = a_lo - 32 * w, hi = a_hi - 32 * w;
            uint32_t keep = 0;
            if (lo <= 31 && hi >= 0 && lo <= hi) {
                const int l = max(lo, 0), h = min(hi, 31);
                keep = (h == 31 ? 0xFFFFFFFFu : ((1u << (h + 1)) - 1u)) & ~((1u << l) - 1u);
            }
            mc[w] &= keep;
        }
    }
#pragma unroll
    for (int w = 0; w < DENSE_MASK_WORDS; w++) {  // grid candidates outside the band (:759-767 / :778-786), ascending d
        if (w >= MW) break;
        // no lane of the wavefront has a candidate in this word (a cell's support disparities sit in one or two of the words): nothing to
        // remove, test or merge - one compare and a scalar branch instead of ~12 instructions (round 5: -4.4 % wave instructions, -2.3 % time at
        // D = 128, -4.4 % at D = 256; the same test on "does a band touch this word" issued MORE, scalar start keys for all-valid wavefronts the same)
        if (__builtin_amdgcn_ballot_w64(mc[w] != 0u) == 0) continue;
        uint32_t m = mc[w] & ~(w == band_word ? band_lo : (w == band_word + 1 ? band_hi : 0u));  // drop [d_plane_min, d_plane_max]
        if (COUNT) ncand += __popc(m);
        int best_w = KEY_NONE;  // keys of this word carry the bit index only; 32 * w is added once per word
        while (m) {  // two candidates per trip: their LDS reads are in flight together (an odd last one is evaluated twice)
            if (COUNT) {  // loop trips: per wavefront (counted by its lowest active lane) and per lane
                const unsigned long long act = __ballot(1);
                npath[3] += ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) ? 1 : 0;
                npath[4]++;
            }
            const int b1 = __ffs((int)m) - 1;
            m &= m - 1;
            const int b2 = m ? __ffs((int)m) - 1 : b1;
            m &= m - 1;  // 0 stays 0
            const uint4 c1 = pu[sgn * (32 * w + b1)], c2 = pu[sgn * (32 * w + b2)];
            best_w = min(best_w, min(sad16_key(own, c1, b1), sad16_key(own, c2, b2)));
        }
        best = min(best, best_w + 32 * w);
    }
    for (int w = DENSE_MASK_WORDS; w < MW; w++) {  // disp_max > 255: remaining words straight from memory
        const uint32_t m = cell[w];
        for (int b = 0; b < 32; b++) {
            const int dc = 32 * w + b;
            if (!((m >> b) & 1u) || dc < a_lo || dc > a_hi || (dc >= d_plane_min && dc <= d_plane_max)) continue;
            if (COUNT) ncand++;
            best = min(best, sad16_key(own, pu[sgn * dc], dc));
        }
    }
    // the band [d_plane - r, d_plane + r], ascending, with the plane prior (:768-774 / :787-793)
    const int r = plane_radius;
    // o' = d - (d_plane - r) in [0, 2r]; the lane's valid candidates are o' in [lo_o, hi_o] (empty when lo_o > hi_o)
    const int lo_o = b_lo - (d_plane - r), hi_o = b_hi - (d_plane - r);
    const int lo_u = __builtin_amdgcn_readfirstlane(lo_o), hi_u = __builtin_amdgcn_readfirstlane(hi_o);
#ifndef DENSE_BAND_FAST
#define DENSE_BAND_FAST 1
#endif
#ifndef DENSE_WAVES_ATTR
#define DENSE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
    if (DENSE_BAND_FAST && r <= 3 && __builtin_amdgcn_ballot_w64(lo_o != lo_u || hi_o != hi_u) == 0) {
        // Fast path (almost every wavefront): all lanes clip the band the same way, so the loop bounds are scalar, the seven
        // candidates are consecutive LDS slots at immediate offsets from one base, and nothing per candidate is left but the
        // read, one v_and_or for the key's start value and the SAD chain.  Keys are relative here (tie-break = o'); the band
        // bit and the band's first disparity are added once at the end (no carry: 0x8000 | d < 0x10000).
        if (lo_u <= hi_u) {
            if (COUNT) ncand += hi_u - lo_u + 1;
            // LDS slot s of the band: pb[s], s = o' (right pixel) or 2r - o' (left pixel: its candidates run towards lower columns)
            const uint4 *pb = side ? pu + (d_plane - r) : pu - (d_plane + r);
            const uint32_t vmask = valid ? 0xFFFFFFFFu : 0x0000FFFFu;  // the prior counts only for a valid plane (:771)
            int bb;
            if (COUNT) npath[(lo_u == 0 && hi_u == 2 * r && (r == 3 || r == 2)) ? 0 : 1]++;
            if (lo_u == 0 && hi_u == 2 * r && r == 3)       // the whole band (the usual case): straight-line code, reads in flight together
                bb = dense_band_full<3>(k, side, own, pb, vmask);
            else if (lo_u == 0 && hi_u == 2 * r && r == 2)
                bb = dense_band_full<2>(k, side, own, pb, vmask);
            else {
                const int p0 = k.prior[0], p1 = k.prior[1], p2 = k.prior[2], p3 = k.prior[3];
                bb = KEY_NONE;
#pragma unroll
                for (int s = 0; s <= 6; s++) {
                    const int o = side ? s : 2 * r - s;                // all scalar
                    if (s > 2 * r || o < lo_u || o > hi_u) continue;
                    const int ao = s < r ? r - s : s - r;              // |o' - r|
                    const int pa = ao == 0 ? p0 : (ao == 1 ? p1 : (ao == 2 ? p2 : p3));
                    const uint32_t start = ((uint32_t)pa << 16) | (uint32_t)o;  // scalar: (prior << 16) | o'
                    bb = min(bb, sad16_key(own, pb[s], (int)(vmask & start)));
                }
            }
            best = min(best, bb + (0x8000 + d_plane - r));
        }
    } else if (DENSE_BAND_FAST && (RT == 3 || RT == 2)) {
        // Lanes clip the band differently (planes near disparity 0 or disp_max, the image's left / right border): all 2r+1 slots
        // are evaluated as in the fast path, and a slot outside the lane's own range starts its key at energy 16384 - above
        // the 10000 a result must beat.  (Such a slot may lie a few entries outside the staged rows: its address is clamped into
        // the buffer, the value is never used.)
        if (COUNT) npath[2]++;
        if (COUNT) ncand += max(hi_o - lo_o + 1, 0);
        const uint4 *pb = side ? pu + (d_plane - r) : pu - (d_plane + r);
        const uint32_t vmask = valid ? 0xFFFFFFFFu : 0x0000FFFFu;
        const int nb = hi_o - lo_o + 1;
        const uint32_t okm = nb > 0 ? ((1u << nb) - 1u) << lo_o : 0u;  // bit o' set: the lane has that candidate
        uint4 c[2 * RT + 1];
#pragma unroll
        for (int q = 0; q <= 2 * RT; q++) c[q] = *min(max(pb + q, lds_first), lds_last);  // (a slot the lane does not have may lie outside the staged rows)
        int bb = KEY_NONE;
#pragma unroll
        for (int q = 0; q <= 2 * RT; q++) {
            const int o = side ? q : 2 * RT - q, ao = q < RT ? RT - q : q - RT;
            // (added, not or-ed: a prior is negative)
            const uint32_t start = (vmask & (((uint32_t)k.prior[ao] << 16) | (uint32_t)o)) + ((__builtin_amdgcn_ubfe(~okm, (uint32_t)o, 1u)) << 30);
            bb = min(bb, sad16_key(own, c[q], (int)start));
        }
        best = min(best, bb + (0x8000 + d_plane - r));  // (a lane without any candidate keeps an energy >= 16375: no result)
    } else {
        // the offset o is uniform over the wavefront, so the prior is a scalar operand and the LDS reads do not depend on earlier iterations
        if (COUNT) npath[2]++;
        for (int o = -r; o <= r; o++) {
            const int dc = d_plane + o;
            if (dc < b_lo || dc > b_hi) continue;
            if (COUNT) ncand++;
            const int prior = valid ? k.prior[o < 0 ? -o : o] : 0;
            best = min(best, sad16_key(own, pu[sgn * dc], prior * 65536 + (0x8000 | dc)));
        }
    }
    return best < (10000 << 16) ? (best & 0x7FFF) : -1;  // :797-800 (min_val starts at 10000, :752); the maps hold these integers as int16
}

// descriptors staged per image: DENSE_TW own columns + disp_max candidates beyond them, rounded so that the right image's
// segment may start up to 3 columns early
__host__ __device__ inline int dense_seg(const Dims &d) { return DENSE_TW + ((d.disp_max + 3) & ~3); }

template <bool COUNT, int MWT, int RT, bool TEX>
__global__ __launch_bounds__(256) DENSE_WAVES_ATTR void k_dense(KParams k, const uint8_t *__restrict__ grad, const int32_t *__restrict__ blob, const int32_t *__restrict__ tri_id,
                                               const float4 *__restrict__ trirec, const uint32_t *__restrict__ gB, int16_t *__restrict__ wta,
                                               unsigned long long *__restrict__ counters) {
    const Dims &d = k.d;
    const int MW = MWT ? MWT : d.MW;
    extern __shared__ uint4 dense_lds[];
    const int pair = blockIdx.z;
    if (blob[pair * META_WORDS] < 3) return;
    const int x0 = blockIdx.x * DENSE_TW, v = d.sub ? 2 * blockIdx.y : blockIdx.y;  // half resolution: even rows only (elas.cpp:919)
    const int x1 = min(x0 + DENSE_TW, d.W);  // tile columns [x0, x1)
    const int yd = max(min(v, d.H - 3), 2);  // elas.cpp:718: descriptor row, clamped (rows 2 and H-3 hold no descriptors: zeros)
    // staged column ranges; both start on a multiple of 4 (x0 is one): descriptors are assembled four columns at a time
    const int l0 = x0, l1 = min(x1 - 1 + d.disp_max, d.W - 1);         // left image  [l0, l1]
    const int r0 = max(x0 - d.disp_max, 0) & ~3, r1 = x1 - 1;          // right image [r0, r1]
    const int qL = (l1 - l0 + 4) >> 2, qR = (r1 - r0 + 4) >> 2;        // quads per image
    uint4 *sL = dense_lds, *sR = dense_lds + dense_seg(d);
    {
        const GradImg gL = grad_image(grad, d, pair, 0), gR = grad_image(grad, d, pair, 1);
        for (int q = threadIdx.x; q < qL + qR; q += 256) {
            if (q < qL)
                expand_quad(gL, d, yd, l0 + 4 * q, sL + 4 * q);
            else
                expand_quad(gR, d, yd, r0 + 4 * (q - qL), sR + 4 * (q - qL));
        }
    }
    // per-pixel global operands of this thread's pixels (two per side), requested before the barrier
    int tt[2][DENSE_TW / 256];
#pragma unroll
    for (int side = 0; side < 2; side++)
#pragma unroll
        for (int j = 0; j < DENSE_TW / 256; j++) {
            const int u = x0 + j * 256 + threadIdx.x;
            tt[side][j] = u < x1 ? tri_id[(size_t)(pair * 2 + side) * d.N + (size_t)v * d.W + u] : -1;
        }
    __syncthreads();
    // grid cell of each of the thread's columns (elas.cpp:745-746), once for both sides; word offsets fit 32 bits.  (Staging the
    // tile's cell masks in LDS instead of one global gather per pixel was measured: 10.4 against 9.2 us per pair.)
    // (the row's cell by the same host-checked multiply-shift as the columns' - scalar - instead of an IEEE division per thread)
    const uint32_t gy = k.cell_mul ? (((uint32_t)v * k.cell_mul) >> 16) : (uint32_t)(int)floorf((float)v / (float)d.grid_size);
    uint32_t cell_off[DENSE_TW / 256];
#pragma unroll
    for (int j = 0; j < DENSE_TW / 256; j++) {
        const int u = min(x0 + j * 256 + (int)threadIdx.x, d.W - 1);
        // floor(u / (float)grid_size) of elas.cpp:745 equals u / grid_size for these integers (the float quotient of a non-multiple is at
        // least 1 / grid_size away from an integer, far more than its rounding error), and (u * cell_mul) >> 16 equals that for every
        // column of the image - the engine checked it (fill_kparams) - or cell_mul is 0 and the float division stays
        const uint32_t cx = k.cell_mul ? (__umul24((uint32_t)u, k.cell_mul) >> 16) : (uint32_t)(int)floorf((float)u / (float)d.grid_size);
        cell_off[j] = gy * (uint32_t)(d.gw * MW) + __umul24(cx, (uint32_t)MW);
    }
    int ncand = 0, npix = 0, npath[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int ps = pair * 2 + side;
#pragma unroll
        for (int j = 0; j < DENSE_TW / 256; j++) {
            const int u = x0 + j * 256 + threadIdx.x;
            if (u >= x1) continue;
            const int t = tt[side][j];
            int out = -10;  // elas.cpp:823-824
            if (t >= 0 && u >= 2 && u < d.W - 2 && !(d.sub && (u & 1))) {
                const float4 rec = trirec[(size_t)ps * d.max_tri + t];
                const uint32_t *cell = gB + (size_t)ps * d.ncell * MW + cell_off[j];
                uint32_t mw[DENSE_MASK_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
                if ((MW & 3) == 0) {  // 16-byte aligned cells: one or two wide loads
                    const uint4 m0 = *reinterpret_cast<const uint4 *>(cell);
                    mw[0] = m0.x, mw[1] = m0.y, mw[2] = m0.z, mw[3] = m0.w;
                    if (MW >= 8) {
                        const uint4 m1 = *reinterpret_cast<const uint4 *>(cell + 4);
                        mw[4] = m1.x, mw[5] = m1.y, mw[6] = m1.z, mw[7] = m1.w;
                    }
                } else {
#pragma unroll
                    for (int w = 0; w < DENSE_MASK_WORDS; w++) mw[w] = w < MW ? cell[w] : 0u;
                }
                const uint4 own = side ? sR[u - r0] : sL[u - l0];
                const uint4 *pu = side ? sL + (u - l0) : sR + (u - r0);  // the other image at the pixel's own column
                out = dense_pixel<COUNT, MWT, RT, TEX>(k, side, u, v, own, pu, rec, mw, cell, ncand, npath, dense_lds, dense_lds + 2 * dense_seg(d) - 1);
                if (COUNT) npix++;
            }
            // integer-valued: a disparity, -1 or -10.  Half resolution (elas.cpp:707-711): only even (u, v) are matched, result at (u/2, v/2)
            if (!d.sub)
                wta[(size_t)ps * d.N + (size_t)v * d.W + u] = (int16_t)out;
            else if (!(u & 1) && (u >> 1) < d.Wm && (v >> 1) < d.Hm)
                wta[(size_t)ps * d.Nm + (size_t)(v >> 1) * d.Wm + (u >> 1)] = (int16_t)out;
        }
    }
    if (COUNT) {
        count_add(counters + CNT_DENSE_CANDIDATES, ncand);
        count_add(counters + CNT_DENSE_PIXELS, npix);
        count_add(counters + CNT_DENSE_BAND_FULL, npath[0]);
        count_add(counters + CNT_DENSE_BAND_PART, npath[1]);
        count_add(counters + CNT_DENSE_BAND_SLOW, npath[2]);
        count_add(counters + CNT_DENSE_GRID_WAVE_TRIPS, npath[3]);
        count_add(counters + CNT_DENSE_GRID_LANE_TRIPS, npath[4]);
    }
}

static size_t dense_lds_bytes(const KParams &k) { return sizeof(uint4) * 2 * (size_t)dense_seg(k.d); }

template <int MWT, int RT, bool TEX>
static void launch_dense_as(const KParams &k, const SlotDev &s, const dim3 &grid, size_t shmem, hipStream_t st) {
    static std::atomic<size_t> granted[64];
    ensure_dynamic_lds(k_dense<false, MWT, RT, TEX>, shmem, granted, "dense_match");
    SV_LAUNCH(K_DENSE, (k_dense<false, MWT, RT, TEX>), grid, dim3(256), shmem, st, k, s.grad, s.blob, s.tri_id, s.trirec, s.gmaskB, s.wta, s.counters);
}

void launch_dense(const KParams &k, const SlotDev &s, int n, hipStream_t st) {
    const size_t shmem = dense_lds_bytes(k);
    const dim3 grid((k.d.W + DENSE_TW - 1) / DENSE_TW, k.d.sub ? (k.d.H + 1) / 2 : k.d.H, n);
    if (s.counters) {
        static std::atomic<size_t> granted_c[64];
        ensure_dynamic_lds(k_dense<true, 0, 0, true>, shmem, granted_c, "dense_match");
        SV_LAUNCH(K_DENSE, (k_dense<true, 0, 0, true>), grid, dim3(256), shmem, st, k, s.grad, s.blob, s.tri_id, s.trirec, s.gmaskB, s.wta, s.counters);
        return;
    }
    // the usual disparity ranges (64 / 128 / 192 / 256) with the presets' plane radii (2: ROBOTICS, 3: MIDDLEBURY) get kernels
    // compiled for them; anything else the generic one
    const int MW = k.d.MW, R = k.plane_radius;
    if (R == 3 && k.match_texture <= 0) {  // MIDDLEBURY / the driver's preset: no texture test
        if (MW == 4) return launch_dense_as<4, 3, false>(k, s, grid, shmem, st);
        if (MW == 8) return launch_dense_as<8, 3, false>(k, s, grid, shmem, st);
        if (MW == 2) return launch_dense_as<2, 3, false>(k, s, grid, shmem, st);
        if (MW == 6) return launch_dense_as<6, 3, false>(k, s, grid, shmem, st);
    }
    if (MW == 4 && R == 3) return launch_dense_as<4, 3, true>(k, s, grid, shmem, st);
    if (MW == 8 && R == 3) return launch_dense_as<8, 3, true>(k, s, grid, shmem, st);
    if (MW == 4 && R == 2) return launch_dense_as<4, 2, true>(k, s, grid, shmem, st);
    if (MW == 8 && R == 2) return launch_dense_as<8, 2, true>(k, s, grid, shmem, st);
    if (MW == 2 && R == 2) return launch_dense_as<2, 2, true>(k, s, grid, shmem, st);
    launch_dense_as<0, 0, true>(k, s, grid, shmem, st);
}

// ------------------------------------------------------------------------------------------------------------
// K6  left/right consistency check      reference: elas.cpp:946-1011
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr(KParams k, const int32_t *__restrict__ blob, const int16_t *__restrict__ wta, float *__restrict__ disp,
                                            float *__restrict__ user_d2, int keep_right) {
    const Dims &d = k.d;
    const int pair = blockIdx.z;
    if (blob[pair * META_WORDS] < 3) return;
    const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (u >= d.W) return;
    const int16_t *W1 = wta + (size_t)(pair * 2) * d.N, *W2 = W1 + d.N;
    const size_t row = (size_t)v * d.W;
    const float d1 = (float)W1[row + u], d2 = (float)W2[row + u];
    const float thr = (float)k.lr_threshold;
    float o1 = -10.0f, o2 = -10.0f;
    // half resolution (elas.cpp:972-975): the map is half size, the disparities are still full-resolution pixels
    const float uw1 = d.sub ? (float)u - d1 / 2 : (float)u - d1, uw2 = d.sub ? (float)u + d2 / 2 : (float)u + d2;
    if (d1 >= 0 && uw1 >= 0 && uw1 < (float)d.W) o1 = (fabsf((float)W2[row + (int)uw1] - d1) > thr) ? -10.0f : d1;
    if (d2 >= 0 && uw2 >= 0 && uw2 < (float)d.W) o2 = (fabsf((float)W1[row + (int)uw2] - d2) > thr) ? -10.0f : d2;
    disp[(size_t)(pair * 2) * d.N + row + u] = o1;
    if (keep_right) disp[(size_t)(pair * 2 + 1) * d.N + row + u] = o2;  // later stages only read it when they process both sides
    if (user_d2) user_d2[(size_t)pair * d.N + row + u] = o2;  // postprocess_only_left: the checked right map is already final
}

// Two pixels per thread for even map widths (every row then starts on a 4-byte boundary of the int16 maps and an 8-byte
// boundary of the float maps): half the load / store instructions of this pure map kernel
template <bool SUB>  // half resolution compiled in: the full-resolution kernel does not compute both warps and select
__global__ __launch_bounds__(256) void k_lr2(KParams k, const int32_t *__restrict__ blob, const int16_t *__restrict__ wta, float *__restrict__ disp,
                                             float *__restrict__ user_d2, int keep_right) {
    const Dims &d = k.d;
    const int pair = blockIdx.z;
    if (blob[pair * META_WORDS] < 3) return;
    const int u = 2 * (blockIdx.x * 256 + threadIdx.x), v = blockIdx.y;
    if (u >= d.W) return;
    const size_t row = (size_t)v * d.W;  // (uniform: the row's pointers are scalar, the columns 32-bit offsets - map_ld)
    const int16_t *W1 = wta + (size_t)(pair * 2) * d.N + row, *W2 = W1 + d.N;
    const short2 a = map_ld(reinterpret_cast<const short2 *>(W1), (uint32_t)u >> 1), b = map_ld(reinterpret_cast<const short2 *>(W2), (uint32_t)u >> 1);
    const float thr = (float)k.lr_threshold;
    float o1[2], o2[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float d1 = (float)(j ? a.y : a.x), d2 = (float)(j ? b.y : b.x);
        const int uj = u + j;
        o1[j] = o2[j] = -10.0f;
        // half resolution (elas.cpp:972-975): the map is half size, the disparities are still full-resolution pixels
        const float uw1 = SUB ? (float)uj - d1 / 2 : (float)uj - d1, uw2 = SUB ? (float)uj + d2 / 2 : (float)uj + d2;
        if (d1 >= 0 && uw1 >= 0 && uw1 < (float)d.W) o1[j] = (fabsf((float)map_ld(W2, (uint32_t)(int)uw1) - d1) > thr) ? -10.0f : d1;
        if (d2 >= 0 && uw2 >= 0 && uw2 < (float)d.W) o2[j] = (fabsf((float)map_ld(W1, (uint32_t)(int)uw2) - d2) > thr) ? -10.0f : d2;
    }
    float *D1r = disp + (size_t)(pair * 2) * d.N + row;
    map_st(reinterpret_cast<float2 *>(D1r), (uint32_t)u >> 1, make_float2(o1[0], o1[1]));
    if (keep_right) map_st(reinterpret_cast<float2 *>(D1r + d.N), (uint32_t)u >> 1, make_float2(o2[0], o2[1]));
    if (user_d2) map_st(reinterpret_cast<float2 *>(user_d2 + (size_t)pair * d.N + row), (uint32_t)u >> 1, make_float2(o2[0], o2[1]));
}

void launch_lr(const KParams &k, const SlotDev &s, int n, hipStream_t st, float *user_d2, bool keep_right) {
    if ((k.d.W & 1) == 0 && (reinterpret_cast<uintptr_t>(user_d2) & 7) == 0 && k.d.sub)
        SV_LAUNCH(K_LR, k_lr2<true>, dim3((k.d.W / 2 + 255) / 256, k.d.H, n), dim3(256), 0, st, k, s.blob, s.wta, s.disp, user_d2, keep_right ? 1 : 0);
    else if ((k.d.W & 1) == 0 && (reinterpret_cast<uintptr_t>(user_d2) & 7) == 0)
        SV_LAUNCH(K_LR, k_lr2<false>, dim3((k.d.W / 2 + 255) / 256, k.d.H, n), dim3(256), 0, st, k, s.blob, s.wta, s.disp, user_d2, keep_right ? 1 : 0);
    else
        SV_LAUNCH(K_LR, k_lr, dim3((k.d.W + 255) / 256, k.d.H, n), dim3(256), 0, st, k, s.blob, s.wta, s.disp, user_d2, keep_right ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------------------
// K7  speckle removal = connected components of valid pixels under |dD| <= speckle_sim_threshold, 4-adjacency
//     reference: elas.cpp:1013-1124 (BFS flood fill).  After the L/R stage every invalid pixel is exactly -10, so the
//     reference's order-dependent BFS reduces to plain component labelling.
//
//     The nodes of the labelling are horizontal RUNS of linked pixels, not pixels:
//       k_ccl_band    one workgroup per band of CCL_R rows.  Reads the band once, keeps three bit masks per row in LDS
//                     (valid / run start / linked to the pixel above), numbers the runs by prefix pop-counts, unions
//                     vertically adjacent runs in an LDS union-find (one thread per 64-pixel mask word, walking its set
//                     bits), and writes one record per run (first pixel, length, root, component size at band-local
//                     roots; records are bump-allocated per map) plus the masks of its first and last row.
//       k_ccl_border / k_ccl_total / k_ccl_apply  ("ccl_finish" in the timing report): three small grid-wide passes over the
//                     run records - unions across band borders (global union-find over run records), component sizes
//                     summed at the global roots, runs of components smaller than speckle_size overwritten with -10.
//                     Only those pixels are written; no per-pixel label map exists.
//     A band with more than ccl_cap runs flags its map; the last workgroup of k_ccl_apply's grid row then labels that map with the slower per-pixel
//     union-find on global memory (ccl_legacy_map, one workgroup per flagged map), so any input is handled.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t map_offset(const Dims &d, int m, int nproc) {  // m = pair*nproc + side
    const int pair = m / nproc, side = m - pair * nproc;
    return (size_t)(pair * 2 + side) * d.N;
}

constexpr int CCL_R = 8;         // rows per band
constexpr int CCL_THREADS_ALONE = 1024;
constexpr int CCL_THREADS = 256;  // (512 until round 5: inside the pipeline an 8-wavefront workgroup with ~55 KB of LDS waited 5 x its own duration for a CU; 256: serial 50 -> 67 us per 32 pairs, pipelined 530 -> 378 us per 64, +0.7 % pairs/s)

// Run records of a map are bump-allocated (a band takes as many as it has runs), `rcap` per map: real disparity maps are far
// more fragmented than smooth synthetic ones (kitti_mini pair 0: 49 000 runs, up to 170 per row), worst case one run per pixel.
struct CclWs {  // workspace views of one launch (ccl_views lays them out)
    int4 *runs;          // [maps][rcap]     (first pixel, length, root as map-wide run index, component size if band-local root)
    int32_t *gparent;    // [maps][rcap]     union-find over the run records
    int32_t *total;      // [maps][rcap]     component size, accumulated at global roots
    int32_t *nruns;      // [maps]           records handed out so far (bump allocator); cleared by k_ccl_apply's last workgroup
    int32_t *boff;       // [maps][nb]       first record of each band
    int32_t *tcount;     // [maps][nb]       runs per band
    int32_t *flag;       // [maps]           == the launch's epoch: some band overflowed (never cleared: the next launch has another epoch)
    uint64_t *bwords;    // [maps][nb][5][nch]  first row: start, valid, up-link masks; last row: start, valid masks
    int32_t *bbase;      // [maps][nb][2][nch]  run number before each 64-pixel word of the first / last row
    int cap, rcap, nb, nch;
};

static size_t ccl_align(size_t x) { return (x + 255) & ~(size_t)255; }

// carves the views out of the workspace; returns the bytes used
static size_t ccl_views(const KParams &k, void *ws, int maps_cap, CclWs &w) {
    w.cap = k.ccl_cap;
    w.rcap = std::max(k.d.Nm / 4, 4096);
    w.nb = (k.d.Hm + CCL_R - 1) / CCL_R;
    w.nch = (k.d.Wm + 63) / 64;
    // sized for the full-resolution map so that one workspace serves both modes of a handle
    const size_t rcap_full = (size_t)std::max(k.d.N / 4, 4096), nb_full = (size_t)(k.d.H + CCL_R - 1) / CCL_R, nch_full = (size_t)(k.d.W + 63) / 64;
    uint8_t *base = static_cast<uint8_t *>(ws);
    size_t o = 0;
    const size_t nodes = (size_t)maps_cap * rcap_full;
    w.runs = reinterpret_cast<int4 *>(base + o);
    o += ccl_align(nodes * sizeof(int4));
    w.gparent = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align(nodes * 4);
    w.total = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align(nodes * 4);
    w.nruns = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align((size_t)maps_cap * 4);
    w.boff = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align((size_t)maps_cap * nb_full * 4);
    w.tcount = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align((size_t)maps_cap * nb_full * 4);
    w.flag = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align((size_t)maps_cap * 4);
    w.bwords = reinterpret_cast<uint64_t *>(base + o);
    o += ccl_align((size_t)maps_cap * nb_full * 5 * nch_full * 8);
    w.bbase = reinterpret_cast<int32_t *>(base + o);
    o += ccl_align((size_t)maps_cap * nb_full * 2 * nch_full * 4);
    return o;
}

size_t ccl_ws_bytes(const KParams &k, int maps_cap) {
    CclWs w;
    return ccl_views(k, nullptr, maps_cap, w);
}

// LDS of k_ccl_band: three bit masks and one run-number prefix per 64-pixel word of the band, 12 B per run
size_t ccl_lds_bytes(const KParams &k) { return (size_t)CCL_R * ((k.d.W + 63) / 64) * (3 * 8 + 4) + (size_t)k.ccl_cap * 2 * 4; }

// union-find with the smaller index as root; works on LDS and on global memory
__device__ __forceinline__ int ccl_find(const int32_t *L, int x) {
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return x;
}

// In the union-find over the run records of a whole map (gparent) the index CCL_LARGE = -1 is a super-root: "belongs to a
// component that is already known to have >= speckle_size pixels".  k_ccl_band points every run of a band-local component
// of that size at it, and since unions hang the larger root under the smaller, anything joined with such a run ends up there
// too.  Only "is the component small?" is ever asked (elas.cpp:1109), so the large components need no identity - and the one
// component that spans a real map (hundreds of band pieces, each large by itself) costs one load per border link instead of
// a walk along a chain of band roots plus contended atomics.
constexpr int CCL_LARGE = -1;

// the same walk on a forest that no longer changes (plain, cacheable loads); CCL_LARGE for members of large components
__device__ __forceinline__ int ccl_find_frozen(const int32_t *L, int x) {
    int p = L[x];
    while (p != x) {
        if (p < 0) return CCL_LARGE;
        x = p;
        p = L[x];
    }
    return x;
}

__device__ __forceinline__ void ccl_union(int32_t *L, int a, int b) {
    for (;;) {
        a = ccl_find(L, a);
        b = ccl_find(L, b);
        if (a == b) return;
        if (a < b) {
            int x = a;
            a = b;
            b = x;
        }
        const int old = atomicMin(&L[a], b);  // a > b: hang the larger root under the smaller
        if (old == a) return;
        a = old;
    }
}

// Variant for the unions across band borders (global memory): a component that spans many bands grows into a chain of band
// roots, and every hop is a dependent L2 access, so the walk halves the path as it goes.  atomicMin keeps the invariant
// "parent <= node and parent is an ancestor": a grandparent is an ancestor for good, concurrent unions only add ancestors.
__device__ __forceinline__ int ccl_find_halving(int32_t *L, int x) {
    if (x < 0) return CCL_LARGE;
    int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        if (p < 0) return CCL_LARGE;
        const int g = __hip_atomic_load(&L[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) atomicMin(&L[x], g);  // g may be CCL_LARGE: x then belongs to the large ones directly
        x = p;
        p = g;
    }
    return x;
}

__device__ __forceinline__ void ccl_union_halving(int32_t *L, int a, int b) {
    for (;;) {
        a = ccl_find_halving(L, a);
        b = ccl_find_halving(L, b);
        if (a == b) return;  // the same component, or both large already
        if (a < b) {
            int x = a;
            a = b;
            b = x;
        }
        const int old = atomicMin(&L[a], b);  // a > b >= CCL_LARGE: hang the larger root under the smaller (or under "large")
        if (old == a) return;
        a = old;
    }
}

__device__ __forceinline__ uint64_t bits_upto(int lane) { return lane == 63 ? ~0ull : ((2ull << lane) - 1ull); }
__device__ __forceinline__ int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }

// Vertical links of one mask word that join two runs not already joined by the link one pixel to the left:
// L = up-link bits, H / Hu = "linked to the left neighbour" bits of this row / the row above, carry = up-link bit of the
// last pixel of the previous word.
__device__ __forceinline__ uint64_t ccl_new_links(uint64_t L, uint64_t H, uint64_t Hu, uint64_t carry) { return L & ~(H & Hu & ((L << 1) | carry)); }

template <int NT>  // CCL_THREADS inside the pipeline; CCL_THREADS_ALONE when the launch is too small to fill the GPU anyway (single pairs): the band's phases are
// chains of dependent loads and barriers, and sixteen wavefronts walk them in a third of the rounds (28 -> ~15 us for one KITTI pair)
__global__ __launch_bounds__(NT) void k_ccl_band(KParams k, int nproc, int epoch, const int32_t *__restrict__ blob, const float *__restrict__ disp, CclWs ws) {
    const Dims &d = k.d;
    extern __shared__ uint64_t ccl_lds[];
    const int m = blockIdx.y, band = blockIdx.x;
    if (blob[(m / nproc) * META_WORDS] < 3) return;
    const int nch = ws.nch, cap = ws.cap;
    const int v0 = band * CCL_R, rows = min(CCL_R, d.H - v0);
    uint64_t *Sm = ccl_lds, *Vm = Sm + CCL_R * nch, *Lm = Vm + CCL_R * nch;  // [CCL_R][nch] run-start / valid / up-link masks
    int32_t *base = reinterpret_cast<int32_t *>(Lm + CCL_R * nch);           // [CCL_R][nch] run starts of the row before the word
    int32_t *parent = base + CCL_R * nch, *len = parent + cap;  // (a run's first pixel goes straight into its record: 16 KB of LDS less than a table of them)
    __shared__ int32_t rowbase[CCL_R + 1];
    __shared__ int32_t rec0;  // first run record of this band
    const float *D = disp + map_offset(d, m, nproc);
    const float thr = k.speckle_sim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NT / 64;
    for (int i = tid; i < cap; i += NT) {
        parent[i] = i;
        len[i] = 0;
    }
    // masks: one wavefront per 64-pixel column strip; the strip's CCL_R+1 rows are requested at once, then reduced to masks
    for (int c = wave; c < nch; c += nw) {
        const int u = c * 64 + lane;
        float val[CCL_R], lft[CCL_R];
        float prev = (v0 > 0 && u < d.W) ? D[(size_t)(v0 - 1) * d.W + u] : -10.0f;
#pragma unroll
        for (int r = 0; r < CCL_R; r++) val[r] = (r < rows && u < d.W) ? D[(size_t)(v0 + r) * d.W + u] : -10.0f;
#pragma unroll
        for (int r = 0; r < CCL_R; r++) lft[r] = (lane == 0 && r < rows && u > 0) ? D[(size_t)(v0 + r) * d.W + u - 1] : -10.0f;
#pragma unroll
        for (int r = 0; r < CCL_R; r++) {
            float left = __shfl_up(val[r], 1, 64);
            if (lane == 0) left = lft[r];
            const bool valid = val[r] >= 0;
            const bool hl = valid && left >= 0 && fabsf(val[r] - left) <= thr;
            const bool vl = valid && prev >= 0 && fabsf(val[r] - prev) <= thr;
            const uint64_t V = __ballot(valid), S = __ballot(valid && !hl), L = __ballot(vl);
            if (lane == 0 && r < rows) {
                Vm[r * nch + c] = V;
                Sm[r * nch + c] = S;
                Lm[r * nch + c] = L;
            }
            prev = val[r];
        }
    }
    __syncthreads();
    // run numbering: exclusive prefix of the start counts along every row, then along the rows
    for (int r = wave; r < rows; r += nw) {
        int run = 0;
        for (int c0 = 0; c0 < nch; c0 += 64) {
            const int c = c0 + lane;
            const int cnt = c < nch ? __popcll(Sm[r * nch + c]) : 0;
            int incl = cnt;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (c < nch) base[r * nch + c] = run + incl - cnt;
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) rowbase[r + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {
        rowbase[0] = 0;
        for (int r = 0; r < rows; r++) rowbase[r + 1] += rowbase[r];
    }
    __syncthreads();
    const int T = rowbase[rows];
    int32_t *tc = ws.tcount + (size_t)m * ws.nb + band;
    if (tid == 0) rec0 = T <= cap ? atomicAdd(&ws.nruns[m], T) : ws.rcap;
    __syncthreads();
    const int off = rec0;
    if (T > cap || off + T > ws.rcap) {  // too many runs for the LDS tables / the map's record pool: the whole map goes the slow way
        if (tid == 0) {
            ws.flag[m] = epoch;
            *tc = 0;
        }
        return;
    }
    // run records (first pixel, length) and vertical unions: one THREAD per quarter of a mask word (16 pixels), walking its set
    // bits.  (One thread per word left 160 of the 512 threads with up to 64 dependent LDS atomics / union-find walks each: 11 of
    // the kernel's 20 us on a real map.)
    int4 *R = ws.runs + (size_t)m * ws.rcap + off;
    for (int task = tid; task < rows * nch * 4; task += NT) {
        const int rc = task >> 2, q = task & 3;
        const uint64_t qmask = 0xFFFFull << (16 * q), qbelow = (1ull << (16 * q)) - 1ull;
        const int r = rc / nch, c = rc - r * nch;
        const uint64_t V = Vm[rc], S = Sm[rc], L = Lm[rc];
        const int b0 = rowbase[r] + base[rc];  // number of the first run that starts in this word
        const int pix0 = (v0 + r) * d.W + c * 64;
        const uint64_t brk = S | ~V;           // a segment ends before the next start or invalid pixel
        if (q == 0 && (V & 1ull) && !(S & 1ull)) {  // the run of the previous word continues into this one
            const uint64_t bk = brk & ~1ull;
            atomicAdd(&len[b0 - 1], bk ? ctz64(bk) : 64);
        }
        int j = b0 + __popcll(S & qbelow);
        for (uint64_t sb = S & qmask; sb; sb &= sb - 1, j++) {
            const int pos = ctz64(sb);
            const uint64_t bk = brk & ~bits_upto(pos);
            atomicAdd(&len[j], (bk ? ctz64(bk) : 64) - pos);
            R[j].x = pix0 + pos;
        }
        if (r >= 1) {
            const uint64_t Vu = Vm[rc - nch], Su = Sm[rc - nch];
            const int bu = rowbase[r - 1] + base[rc - nch];
            const uint64_t carry = c > 0 ? (Lm[rc - 1] >> 63) : 0ull;
            for (uint64_t F = ccl_new_links(L, V & ~S, Vu & ~Su, carry) & qmask; F; F &= F - 1) {
                const uint64_t upto = bits_upto(ctz64(F));
                ccl_union(parent, b0 + __popcll(S & upto) - 1, bu + __popcll(Su & upto) - 1);
            }
        }
    }
    __syncthreads();
    // flatten; the records take (first pixel, length, root); then the lengths of the non-roots are folded into their root's slot
    // (a non-root's own slot is never a target), which leaves the component sizes at the roots
    int32_t *GP = ws.gparent + (size_t)m * ws.rcap + off, *TOT = ws.total + (size_t)m * ws.rcap + off;
    for (int i = tid; i < T; i += NT) {
        const int root = ccl_find(parent, i);
        parent[i] = root;
        R[i].y = len[i], R[i].z = off + root, R[i].w = 0;
        GP[i] = off + root;
        TOT[i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < T; i += NT)
        if (parent[i] != i) atomicAdd(&len[parent[i]], len[i]);
    __syncthreads();
    for (int i = tid; i < T; i += NT) {
        if (parent[i] == i) R[i].w = len[i];
        if (len[parent[i]] >= k.speckle_size) GP[i] = CCL_LARGE;  // a band-local component of that size: large whatever it joins
    }
    if (tid == 0) {
        *tc = T;
        ws.boff[(size_t)m * ws.nb + band] = off;
    }
    uint64_t *bw = ws.bwords + ((size_t)m * ws.nb + band) * 5 * nch;
    int32_t *bb = ws.bbase + ((size_t)m * ws.nb + band) * 2 * nch;
    for (int c = tid; c < nch; c += NT) {
        const int rl = (rows - 1) * nch + c;
        bw[c] = Sm[c];
        bw[nch + c] = Vm[c];
        bw[2 * nch + c] = Lm[c];
        bw[3 * nch + c] = Sm[rl];
        bw[4 * nch + c] = Vm[rl];
        bb[c] = base[c];
        bb[nch + c] = rowbase[rows - 1] + base[rl];
    }
}

// ---- per-pixel union-find on global memory: the slow path for maps whose bands overflow the run tables ----
// csize[p]: run length (> 0) at the first pixel of every horizontal run of linked valid pixels, 0 elsewhere.
// cnt[p]  : component size accumulator, meaningful at root pixels (roots are always run starts).
__device__ void ccl_legacy_init_row(const KParams &k, const float *D, int32_t *L, int32_t *S, int32_t *C, int v, int lane) {
    const Dims &d = k.d;
    int carry_start = -1;
    float carry_val = -10.0f;
    for (int u0 = 0; u0 < d.W; u0 += 64) {
        const int u = u0 + lane;
        const float val = u < d.W ? D[u] : -10.0f;
        const bool valid = val >= 0;
        float left = __shfl_up(val, 1, 64);
        if (lane == 0) left = carry_val;
        const bool link = valid && left >= 0 && fabsf(val - left) <= k.speckle_sim;
        const unsigned long long brk = __ballot(!link);  // bit set: this pixel starts a run (or is invalid)
        const unsigned long long upto = brk & bits_upto(lane);
        const int start = upto ? (u0 + 63 - __clzll((long long)upto)) : carry_start;
        // the run ends here if the next pixel does not link to this one (next chunk's first pixel: decided there)
        const bool next_breaks = lane == 63 ? false : ((brk >> (lane + 1)) & 1ull) != 0;
        if (u < d.W) {
            L[u] = valid ? (int)(v * d.W + start) : -1;
            if (!(valid && start == u)) S[u] = 0;       // run starts get their length from the run's last pixel
            C[u] = 0;
            if (valid && (next_breaks || u == d.W - 1)) S[start] = u - start + 1;
        }
        // a run that reaches lane 63 continues into the next chunk unless that chunk's lane 0 breaks it
        carry_start = __shfl(valid ? start : -1, 63, 64);
        carry_val = __shfl(val, 63, 64);
        if (u0 + 64 < d.W) {
            // peek: does the first pixel of the next chunk link to our lane 63?  If not, close the run now.
            const float nxt = D[u0 + 64];
            const bool cont = carry_start >= 0 && nxt >= 0 && fabsf(nxt - carry_val) <= k.speckle_sim;
            if (lane == 63 && valid && !cont) S[start] = u - start + 1;
        }
    }
}

__device__ __forceinline__ void ccl_phase_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __syncthreads();
}

__device__ void ccl_legacy_map(const KParams &k, float *D, int32_t *label, int32_t *csize, int32_t *cnt) {
    const Dims &d = k.d;
    const int tid = threadIdx.x, nt = blockDim.x;
    const float thr = k.speckle_sim;
    for (int v = tid >> 6; v < d.H; v += nt >> 6)
        ccl_legacy_init_row(k, D + (size_t)v * d.W, label + (size_t)v * d.W, csize + (size_t)v * d.W, cnt + (size_t)v * d.W, v, tid & 63);
    ccl_phase_sync();
    for (int p = d.W + tid; p < d.N; p += nt) {  // vertical links
        const int u = p % d.W;
        const float c = D[p], up = D[p - d.W];
        if (!(c >= 0 && up >= 0 && fabsf(c - up) <= thr)) continue;
        if (u > 0) {  // the same vertical link already exists one pixel to the left inside both runs: nothing new to merge
            const float cl = D[p - 1], ul = D[p - d.W - 1];
            if (cl >= 0 && ul >= 0 && fabsf(c - cl) <= thr && fabsf(up - ul) <= thr && fabsf(cl - ul) <= thr) continue;
        }
        ccl_union(label, p, p - d.W);
    }
    ccl_phase_sync();
    for (int p = tid; p < d.N; p += nt) {  // one root chase + add per run
        const int len = csize[p];
        if (len <= 0) continue;
        const int root = ccl_find(label, p);
        if (root != p) label[p] = root;
        atomicAdd(&cnt[root], len);
    }
    ccl_phase_sync();
    for (int p = tid; p < d.N; p += nt) {
        const int start = __hip_atomic_load(&label[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (start < 0) continue;
        const int root = __hip_atomic_load(&label[start], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // run starts were compressed above
        if (__hip_atomic_load(&cnt[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k.speckle_size) D[p] = -10.0f;  // elas.cpp:1109-1114
    }
}

// ---- second half of the run-based labelling: three small grid-wide passes over the run records ----
// (a) unions across band borders: first row of band b against the last row of band b-1.  One WAVEFRONT per 64-pixel mask word,
// one lane per pixel: every union is a chain of dependent L2 accesses, so the links of a word are joined side by side, not one
// after the other (a word of a real map has up to ten of them)
__global__ __launch_bounds__(256) void k_ccl_border(int nproc, int epoch, const int32_t *__restrict__ blob, CclWs ws) {
    const int m = blockIdx.y;
    if (blob[(m / nproc) * META_WORDS] < 3 || ws.flag[m] == epoch) return;
    const int nch = ws.nch, nb = ws.nb;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave-uniform word index
    if (idx >= (nb - 1) * nch) return;
    const int b = 1 + idx / nch, c = idx - (b - 1) * nch;
    const uint64_t *bwb = ws.bwords + ((size_t)m * nb + b) * 5 * nch, *bwp = bwb - 5 * nch;
    const uint64_t L = bwb[2 * nch + c];
    if (!L) return;
    const uint64_t S0 = bwb[c], V0 = bwb[nch + c], Sp = bwp[3 * nch + c], Vp = bwp[4 * nch + c];
    const uint64_t carry = c > 0 ? (bwb[2 * nch + c - 1] >> 63) : 0ull;
    const uint64_t F = ccl_new_links(L, V0 & ~S0, Vp & ~Sp, carry);
    if (!((F >> lane) & 1ull)) return;
    const int32_t *bbb = ws.bbase + ((size_t)m * nb + b) * 2 * nch, *bbp = bbb - 2 * nch;
    const int32_t *BO = ws.boff + (size_t)m * nb;
    const int n0 = BO[b] + bbb[c], np = BO[b - 1] + bbp[nch + c];
    int32_t *GP = ws.gparent + (size_t)m * ws.rcap;
    const uint64_t upto = bits_upto(lane);
    ccl_union_halving(GP, n0 + __popcll(S0 & upto) - 1, np + __popcll(Sp & upto) - 1);
}

// (b) component sizes at the global roots: every band-local root adds its component's pixel count.  CCL_SPLIT workgroups per
// band: the walk to a root is a chain of dependent loads, so a thread should not have to do more than one or two of them
constexpr int CCL_SPLIT = 4;
__global__ __launch_bounds__(256) void k_ccl_total(int nproc, int epoch, int speckle_size, const int32_t *__restrict__ blob, CclWs ws) {
    const int m = blockIdx.y, b = blockIdx.x / CCL_SPLIT, part = blockIdx.x - b * CCL_SPLIT;
    if (blob[(m / nproc) * META_WORDS] < 3 || ws.flag[m] == epoch) return;
    const int T = ws.tcount[(size_t)m * ws.nb + b], off = ws.boff[(size_t)m * ws.nb + b];
    int32_t *GP = ws.gparent + (size_t)m * ws.rcap;
    int32_t *TOT = ws.total + (size_t)m * ws.rcap;
    const int4 *RUNS = ws.runs + (size_t)m * ws.rcap;
    for (int i = part * 256 + threadIdx.x; i < T; i += 256 * CCL_SPLIT) {
        const int4 r = RUNS[off + i];
        if (r.z != off + i || r.w >= speckle_size) continue;  // band-local roots of small components only (the others are CCL_LARGE)
        const int root = ccl_find_frozen(GP, off + i);
        if (root < 0) {  // joined a large component across some border
            GP[off + i] = CCL_LARGE;
            continue;
        }
        // Only "total >= speckle_size" is ever asked (k_ccl_apply) and totals only grow: once a root has reached the threshold
        // further adds are skipped
        if (__hip_atomic_load(&TOT[root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < speckle_size) atomicAdd(&TOT[root], r.w);
        // a component made of many small band pieces is a chain of band roots: point this one at the global root so that
        // k_ccl_apply gets there in one hop (no union runs any more; a concurrent walk sees the old parent or the root, both ancestors)
        GP[off + i] = root;
    }
}

// (c) runs of components smaller than speckle_size are wiped (elas.cpp:1109-1114); nothing else is written.
// (d) The LAST workgroup of every map's row of the grid (blockIdx.x == nb * CCL_SPLIT) is the slow path for maps whose bands overflowed
// the run tables: per-pixel union-find, one workgroup per flagged map (normally none - it returns at once).  It used to be a launch of its
// own (one more 5 us link in a single pair's chain, and up to 78 us per chunk beside the other streams' kernels); that the two can share a
// launch is what the epoch is for: a band marks its map with the number of the launch, nobody has to clear the mark while others still read it.
__global__ __launch_bounds__(256) void k_ccl_apply(KParams k, int nproc, int epoch, const int32_t *__restrict__ blob, float *__restrict__ disp, CclWs ws, int32_t *__restrict__ label,
                                                   int32_t *__restrict__ csize, int32_t *__restrict__ cnt) {
    const Dims &d = k.d;
    const int m = blockIdx.y;
    if ((int)blockIdx.x == ws.nb * CCL_SPLIT) {
        if (threadIdx.x == 0) ws.nruns[m] = 0;  // the record pool is free again for the slot's next launch (nobody allocates after k_ccl_band)
        if (blob[(m / nproc) * META_WORDS] < 3 || ws.flag[m] != epoch) return;
        const size_t off = map_offset(d, m, nproc);
        ccl_legacy_map(k, disp + off, label + off, csize + off, cnt + off);
        return;
    }
    const int b = blockIdx.x / CCL_SPLIT, part = blockIdx.x - b * CCL_SPLIT;
    if (blob[(m / nproc) * META_WORDS] < 3 || ws.flag[m] == epoch) return;
    const int T = ws.tcount[(size_t)m * ws.nb + b], off = ws.boff[(size_t)m * ws.nb + b];
    const int32_t *GP = ws.gparent + (size_t)m * ws.rcap, *TOT = ws.total + (size_t)m * ws.rcap;
    const int4 *RUNS = ws.runs + (size_t)m * ws.rcap;
    float *D = disp + map_offset(d, m, nproc);
    for (int i = part * 256 + threadIdx.x; i < T; i += 256 * CCL_SPLIT) {
        const int4 r = RUNS[off + i];
        if (r.y >= k.speckle_size) continue;  // a run that long is a large component by itself
        const int root = ccl_find_frozen(GP, off + i);  // (its own entry: CCL_LARGE at once for runs of large band components)
        if (root < 0 || TOT[root] >= k.speckle_size) continue;
        for (int q = 0; q < r.y; q++) D[r.x + q] = -10.0f;
    }
}

void launch_speckle(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st) {
    const int maps = n * nproc;
    CclWs ws;
    ccl_views(k, s.ccl_ws, s.cap * 2, ws);
    const size_t lds = ccl_lds_bytes(k);
    const int epoch = (int)(++s.ccl_epoch & 0x3FFFFFFFu) + 1;  // > 0: the workspace's marks start cleared
    static std::atomic<size_t> granted[64], granted_alone[64];
    if ((size_t)ws.nb * maps <= 256) {  // less than one workgroup per CU: the bands' own latency is all there is
        ensure_dynamic_lds(k_ccl_band<CCL_THREADS_ALONE>, lds, granted_alone, "ccl_band");
        SV_LAUNCH(K_CCL_BAND, k_ccl_band<CCL_THREADS_ALONE>, dim3(ws.nb, maps), dim3(CCL_THREADS_ALONE), lds, st, k, nproc, epoch, s.blob, s.disp, ws);
    } else {
        ensure_dynamic_lds(k_ccl_band<CCL_THREADS>, lds, granted, "ccl_band");  // wide images: larger run tables than the default dynamic-LDS limit allows
        SV_LAUNCH(K_CCL_BAND, k_ccl_band<CCL_THREADS>, dim3(ws.nb, maps), dim3(CCL_THREADS), lds, st, k, nproc, epoch, s.blob, s.disp, ws);
    }
    if (ws.nb > 1) SV_LAUNCH(K_CCL_FINISH, k_ccl_border, dim3(((ws.nb - 1) * ws.nch + 3) / 4, maps), dim3(256), 0, st, nproc, epoch, s.blob, ws);
    SV_LAUNCH(K_CCL_FINISH, k_ccl_total, dim3(ws.nb * CCL_SPLIT, maps), dim3(256), 0, st, nproc, epoch, k.speckle_size, s.blob, ws);
    int32_t *cnt = reinterpret_cast<int32_t *>(s.tmp);  // slow path only: the filters' scratch map is free during speckle removal
    SV_LAUNCH(K_CCL_FINISH, k_ccl_apply, dim3(ws.nb * CCL_SPLIT + 1, maps), dim3(256), 0, st, k, nproc, epoch, s.blob, s.disp, ws, s.tri_id, s.csize, cnt);
}

// ------------------------------------------------------------------------------------------------------------
// K8  gap interpolation, rows then columns      reference: elas.cpp:1126-1294
//     Filled pixels never serve as interpolation end points (those are always original valid pixels), so each
//     line can be resolved from its validity bit mask: previous / next valid position by bit scans.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gap_value(float d1, float d2) {
    return fabsf(d1 - d2) < 3.0f ? (d1 + d2) / 2 : (d2 < d1 ? d2 : d1);  // :1168-1171 (discon_threshold 3.0)
}

#define SV_MAX_LINE_WORDS 128  // lines up to 8192 pixels

// One workgroup (4 wavefronts) per row: the wavefronts share the row's 64-pixel words, so a row is a chain of nch / 4 memory
// latencies instead of nch (measured inside the one-wavefront version at batch 1: 5.8 us of dependent loads, 2.2 us of serial
// scan by lane 0, 6.0 us of resolve for a 1242-pixel row)
#define GAPR_THREADS 256
__global__ __launch_bounds__(GAPR_THREADS) void k_gap_rows(KParams k, int nproc, const int32_t *__restrict__ blob, float *__restrict__ disp) {
    const Dims &d = k.d;
    const int m = blockIdx.y;
    if (blob[(m / nproc) * META_WORDS] < 3) return;
    float *D = disp + map_offset(d, m, nproc) + (size_t)blockIdx.x * d.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = GAPR_THREADS / 64;
    const int nch = (d.W + 63) / 64;
    __shared__ unsigned long long mw[SV_MAX_LINE_WORDS];
    __shared__ int prevb[SV_MAX_LINE_WORDS], nextb[SV_MAX_LINE_WORDS];
    for (int c = wave; c < nch; c += nw) {
        const int u = c * 64 + lane;
        const float val = u < d.W ? map_ld(D, (uint32_t)u) : -1.0f;
        const unsigned long long b = __ballot(val >= 0);
        if (lane == 0) mw[c] = b;
    }
    __syncthreads();
    // last valid pixel before / first valid pixel after each word: one lane per word, max / min scans across the wavefront
    if (wave == 0) {
        int carry = -1;
        for (int c0 = 0; c0 < nch; c0 += 64) {
            const int c = c0 + lane;
            const unsigned long long w = c < nch ? mw[c] : 0ull;
            int last = w ? c * 64 + 63 - __clzll((long long)w) : -1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(last, o, 64);
                if (lane >= o) last = max(last, t);
            }
            int excl = __shfl_up(last, 1, 64);
            if (lane == 0) excl = -1;
            if (c < nch) prevb[c] = max(excl, carry);
            carry = max(carry, __shfl(last, 63, 64));
        }
    } else if (wave == 1) {
        int carry = 0x7FFFFFFF;
        for (int c0 = ((nch - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
            const int c = c0 + lane;
            const unsigned long long w = c < nch ? mw[c] : 0ull;
            int first = w ? c * 64 + __ffsll((long long)w) - 1 : 0x7FFFFFFF;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(first, o, 64);
                if (lane + o < 64) first = min(first, t);
            }
            int excl = __shfl_down(first, 1, 64);
            if (lane == 63) excl = 0x7FFFFFFF;
            const int nb = min(excl, carry);
            if (c < nch) nextb[c] = nb == 0x7FFFFFFF ? -1 : nb;
            carry = min(carry, __shfl(first, 0, 64));
        }
    }
    __syncthreads();
    const int gw = k.gap_width;
    for (int c = wave; c < nch; c += nw) {
        const int u = c * 64 + lane;
        const unsigned long long w = mw[c];
        if (u >= d.W || ((w >> lane) & 1ull)) continue;
        const unsigned long long below = w & ((1ull << lane) - 1ull);
        const unsigned long long above = lane == 63 ? 0ull : (w >> (lane + 1));
        const int pl = below ? c * 64 + 63 - __clzll((long long)below) : prevb[c];
        const int nr = above ? u + __ffsll((long long)above) : nextb[c];
        if (pl >= 0 && nr >= 0) {
            if (nr - pl - 1 <= gw) map_st(D, (uint32_t)u, gap_value(map_ld(D, (uint32_t)pl), map_ld(D, (uint32_t)nr)));  // :1158-1176
        } else if (k.add_corners) {
            if (pl < 0 && nr >= 0) {
                if (u >= nr - gw) map_st(D, (uint32_t)u, map_ld(D, (uint32_t)nr));  // :1191-1201
            } else if (nr < 0 && pl >= 0) {
                if (u <= pl + gw) map_st(D, (uint32_t)u, map_ld(D, (uint32_t)pl));  // :1204-1214
            }
        }
    }
}

void launch_gap_rows(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st) {
    SV_LAUNCH(K_GAP_ROWS, k_gap_rows, dim3(k.d.H, n * nproc), dim3(GAPR_THREADS), 0, st, k, nproc, s.blob, s.disp);
}

// Columns: a workgroup owns 64 columns; its 4 wavefronts split the rows.  Pass 1 builds the per-column validity
// bit masks in LDS (coalesced row reads), pass 2 resolves every invalid pixel on its own from the masks.
#define GAPC_THREADS 256  // a wavefront takes 64-row words of the column masks in turns (512 threads until round 5: pipelined 166 -> 62 us per 64-pair launch)
__global__ __launch_bounds__(GAPC_THREADS) void k_gap_cols(KParams k, int nproc, const int32_t *__restrict__ blob, float *__restrict__ disp) {
    const Dims &d = k.d;
    const int m = blockIdx.y;
    if (blob[(m / nproc) * META_WORDS] < 3) return;
    float *D = disp + map_offset(d, m, nproc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * 64 + lane;
    extern __shared__ unsigned long long cmask[];  // [nw][64]
    const int nw = (d.H + 63) / 64;
    const bool live = u < d.W;
    for (int w = wave; w < nw; w += GAPC_THREADS / 64) {
        unsigned long long word = 0;
        const int vend = min(64, d.H - w * 64);
        const float *col = D + (size_t)(w * 64) * d.W + (live ? u : 0);
        for (int b0 = 0; b0 < vend; b0 += 16) {  // sixteen rows in flight at a time (one load per trip was a chain of 64 latencies: 17.6 us)
            float val[16];
#pragma unroll
            for (int j = 0; j < 16; j++) val[j] = col[(size_t)min(b0 + j, vend - 1) * d.W];
#pragma unroll
            for (int j = 0; j < 16; j++) word |= (unsigned long long)(live && b0 + j < vend && val[j] >= 0) << (b0 + j);
        }
        cmask[w * 64 + lane] = word;
    }
    __syncthreads();
    if (!live) return;
    const int gw = k.gap_width;
    for (int w = wave; w < nw; w += GAPC_THREADS / 64) {
        const unsigned long long word = cmask[w * 64 + lane];
        const int vend = min(64, d.H - w * 64);
        unsigned long long inval = ~word & (vend == 64 ? ~0ull : ((1ull << vend) - 1ull));
        while (inval) {
            const int b = __ffsll((long long)inval) - 1;
            inval &= inval - 1;
            const int v = w * 64 + b;
            // previous valid row
            int pv = -1;
            {
                unsigned long long below = word & ((1ull << b) - 1ull);
                int ww = w;
                for (;;) {
                    if (below) {
                        pv = ww * 64 + 63 - __clzll((long long)below);
                        break;
                    }
                    if (--ww < 0) break;
                    below = cmask[ww * 64 + lane];
                }
            }
            int nv = -1;
            {
                unsigned long long above = b == 63 ? 0ull : (word >> (b + 1)) << (b + 1);
                int ww = w;
                for (;;) {
                    if (above) {
                        nv = ww * 64 + __ffsll((long long)above) - 1;
                        break;
                    }
                    if (++ww >= nw) break;
                    above = cmask[ww * 64 + lane];
                }
            }
            if (pv >= 0 && nv >= 0) {
                if (nv - pv - 1 <= gw) D[(size_t)v * d.W + u] = gap_value(D[(size_t)pv * d.W + u], D[(size_t)nv * d.W + u]);  // :1232-1251
            } else if (k.add_corners) {
                if (pv < 0 && nv >= 0) {
                    if (v >= nv - gw) D[(size_t)v * d.W + u] = D[(size_t)nv * d.W + u];  // :1268-1278
                } else if (nv < 0 && pv >= 0) {
                    if (v <= pv + gw) D[(size_t)v * d.W + u] = D[(size_t)pv * d.W + u];  // :1281-1291
                }
            }
        }
    }
}

void launch_gap_cols(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st) {
    const size_t shmem = (size_t)((k.d.H + 63) / 64) * 64 * sizeof(unsigned long long);
    SV_LAUNCH(K_GAP_COLS, k_gap_cols, dim3((k.d.W + 63) / 64, n * nproc), dim3(GAPC_THREADS), shmem, st, k, nproc, s.blob, s.disp);
}

// ------------------------------------------------------------------------------------------------------------
// K9  adaptive mean (8-tap "bilateral" approximation), horizontal then vertical   reference: elas.cpp:1297-1494
//     Bit-exactness needs (a) the reference's broken absolute-value mask (:1329, _mm_set1_ps(0x7FFFFFFF) is the
//     float 2^31 = 0x4F000000) and (b) its summation order: ring slot = pixel index mod 8, SSE lanes pair slot j
//     with j+4, then ((s0+s1)+s2)+s3 (:1427-1434).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float absq(float x) { return __uint_as_float(__float_as_uint(x) & 0x4F000000u); }

// factor_sum / weight_sum.  A weight is 4, 2 or 0 (absq() leaves 2^odd or something below 2^-96), so weight_sum is an even
// integer in [2, 32]; for those sixteen divisors  y = v_rcp_f32(d), q = a*y, q + fma(-q, d, a)*y  is the correctly rounded
// quotient for EVERY float a - k_check_amean_div compares it with the IEEE division (the 10-instruction v_div_scale /
// v_div_fmas / v_div_fixup sequence) over all 2^23 mantissas, both signs, 26 exponents and the sixteen divisors on the GPU the
// engine runs on (tests/test_gpu_parity.py::test_adaptive_mean_division_is_exact).
__device__ __forceinline__ float amean_div(float a, float d) {
    const float y = __builtin_amdgcn_rcpf(d);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q, d, a), y, q);
}

// mism[0]: number of (a, d) pairs where amean_div differs from a / d; mism[1..2]: bits of the first such a and d;
// mism[3]: the same count for the uncorrected a * rcp(d) - the control that shows the comparison can fail
__global__ __launch_bounds__(256) void k_check_amean_div(unsigned long long *mism) {
    const uint32_t mant = blockIdx.x * 256u + threadIdx.x;  // 2^23 mantissas
    unsigned long long bad = 0, bad_plain = 0;
    uint32_t first_a = 0, first_d = 0;
    for (int e = 112; e <= 142; e++)                         // a in [2^-15, 2^16)
        for (int sgn = 0; sgn < 2; sgn++) {
            const float a = __uint_as_float(((uint32_t)sgn << 31) | ((uint32_t)e << 23) | mant);
            for (int di = 1; di <= 32; di++) {  // 2, 4, ..., 32 (weights 4 / 2 / 0: k_amean_sub) and 0.5, 1, ..., 8 (weights 1 / 0.5 / 0: k_amean)
                const float d = di <= 16 ? (float)(2 * di) : 0.5f * (float)(di - 16);
                const uint32_t want = __float_as_uint(a / d);
                if (want != __float_as_uint(amean_div(a, d))) {
                    if (bad == 0) first_a = __float_as_uint(a), first_d = __float_as_uint(d);
                    bad++;
                }
                bad_plain += want != __float_as_uint(a * __builtin_amdgcn_rcpf(d)) ? 1 : 0;
            }
        }
    if (mant == 0)  // a = 0
        for (int di = 1; di <= 32; di++)
            if (__float_as_uint(amean_div(0.0f, di <= 16 ? (float)(2 * di) : 0.5f * (float)(di - 16))) != 0u) bad++;
    if (bad && atomicAdd(mism, bad) == 0) {
        mism[1] = first_a;
        mism[2] = first_d;
    }
    if (bad_plain) atomicAdd(mism + 3, bad_plain);
}

int launch_check_amean_div(unsigned long long *d_mism, hipStream_t st) {
    hipLaunchKernelGGL(k_check_amean_div, dim3((1u << 23) / 256), dim3(256), 0, st, d_mism);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- full resolution: several outputs per thread along the pass direction ---------------------------------------------------------
// The weight of a tap depends on |tap - centre| only through the masked bits (absq), and a - b and b - a differ in the sign bit alone,
// which the mask drops: w(a, b) = w(b, a).  A thread therefore takes EIGHT consecutive centres of a line: their windows (centre-4 ..
// centre+3) cover 15 values, 38 distinct pairs instead of 8 x 7 = 56 taps, and the 15 values come from LDS once instead of 72 times.
// Weights are kept divided by four - 1, 0.5 or 0: one v_fma_f32 with the clamp modifier, fma(absq(a - b), -0.25, 1) clamped to [0, 1],
// gives max(4 - absq, 0) / 4 exactly (absq is 2^odd >= 2 or below 2^-96) - so every product x * w is exact, every partial sum is the
// reference's partial sum divided by four (scaling by a power of two commutes with rounding), and the quotient is the same real number:
// same bits.  The centre's own tap has weight 1, so weight_sum >= 1: the reference's weight_sum > 0 test (:1436) cannot fail.
// Ring slot of a pixel = its index mod 8 (:1404-1411); a run starts at a multiple of 4, so the slot of value i is static
// whatever the centre, and so is the lane-sum order (:1427-1434) - slots (0+4) + (1+5) + (2+6) + (3+7).
__device__ __forceinline__ float amean_w4(float a, float b, float minus_quarter) {  // weight / 4 of a tap against a centre (symmetric)
    const float q = absq(a - b);
    float w;
    asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(w) : "v"(q), "v"(minus_quarter));
    return w;
}

// v[0 .. NC + 6]: the line's values at positions P-4 .. P+NC+2 (P a multiple of 4); centre c (0..NC-1) is v[c + 4].  res[c] / ok[c]: the
// filtered value and whether the reference stores it (the quotient is >= 0, :1437).  Ring slots: value i sits in slot (i + 4) & 7 when P
// is a multiple of 8 and in slot i & 7 otherwise - slots j and j + 4 trade places, and the lane sums add exactly those two first.
template <int NC>
__device__ __forceinline__ void amean_run(const float (&v)[NC + 7], float (&res)[NC], bool (&ok)[NC]) {
    const float mq = -0.25f;
    float wt[NC + 7][4];  // wt[a][d - 1] = weight between v[a] and v[a + d]; only the pairs some centre needs exist after unrolling
#pragma unroll
    for (int a = 0; a < NC + 7; a++)
#pragma unroll
        for (int dd = 1; dd <= 4; dd++) {
            const int b = a + dd;
            const bool need = b <= NC + 6 && ((a >= 4 && a <= NC + 3 && dd <= 3) || (b >= 4 && b <= NC + 3));  // a is a centre with tap +dd, or b one with tap -dd
            wt[a][dd - 1] = need ? amean_w4(v[a], v[b], mq) : 0.0f;
        }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int ctr = c + 4;
        float x[8], w[8];
        bool self[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {  // ring slot j of this centre's window
            const int i = c + ((j - 4 - c) & 7);
            self[j] = i == ctr;
            x[j] = v[i];
            w[j] = i == ctr ? 1.0f : (i < ctr ? wt[i][ctr - i - 1] : wt[ctr][i - ctr - 1]);
        }
        float p[4], ws[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // f[k] + f[k + 4] with f = x * w exact: one rounding, in the fma
            if (self[k])
                p[k] = __builtin_fmaf(x[k + 4], w[k + 4], x[k]);
            else if (self[k + 4])
                p[k] = __builtin_fmaf(x[k], w[k], x[k + 4]);
            else
                p[k] = __builtin_fmaf(x[k + 4], w[k + 4], x[k] * w[k]);
            ws[k] = w[k] + w[k + 4];
        }
        const float weight_sum = ((ws[0] + ws[1]) + ws[2]) + ws[3];  // elas.cpp:1427-1434 (multiples of 0.5 up to 8: exact in any order)
        const float factor_sum = ((p[0] + p[1]) + p[2]) + p[3];
        const float dd = amean_div(factor_sum, weight_sum);
        res[c] = dd;
        ok[c] = dd >= 0;
    }
}

// Both passes in one kernel: a 64 x 32 output tile needs the horizontal result on rows y-4..y+3, which needs the input on
// columns x-4..x+3; both live in LDS, so every input value is fetched from memory ~1.4 times instead of 16.  Out of place
// (src -> dst): the vertical pass of another workgroup must never see this one's output.  A task = AM_NC consecutive centres of a line.
#define AM_TW 64
#define AM_TH 32
#define AM_NC 8  // (4: 861 k wave-instructions per pair, 8: 789 k at 80 VGPRs and the same 1.7 us; 8 with 24-row tiles: 821 k)
#define PF_TW 64
#define PF_TH 32

__global__ __launch_bounds__(256) void k_amean(KParams k, int nproc, const int32_t *__restrict__ blob, const float *__restrict__ src, float *__restrict__ dst) {
    const Dims &d = k.d;
    const int m = blockIdx.z;
    if (blob[(m / nproc) * META_WORDS] < 3) return;
    const size_t off = map_offset(d, m, nproc);
    const float *S = src + off;
    const int x0 = blockIdx.x * AM_TW, y0 = blockIdx.y * AM_TH;
    constexpr int ROWS = AM_TH + 7, DCOLS = AM_TW + 8, NV = AM_NC + 7;
    __shared__ __attribute__((aligned(16))) float sD[ROWS][DCOLS];  // D_copy (:1307-1318): rows y0-4.., columns x0-4..; invalid -> -10
    __shared__ __attribute__((aligned(16))) float sT[ROWS][AM_TW];  // D_tmp after the horizontal pass: rows y0-4.., columns x0..
    {  // all of a thread's tile loads are requested before the first one is used (one load per loop trip is a chain of latencies).
       // A thread owns one column PAIR of the tile (x0 - 4 is even) and every 7th row: its column tests and offsets are computed once.
        constexpr int PAIRS = DCOLS / 2, RPP = 256 / PAIRS, NLD = (ROWS + RPP - 1) / RPP;  // 36 pairs, 7 rows per pass, 6 passes
        const int cp = threadIdx.x % PAIRS, rp = threadIdx.x / PAIRS;
        const int x = x0 - 4 + 2 * cp;
        const bool live = rp < RPP, in0 = x >= 0 && x < d.W, in1 = x + 1 >= 0 && x + 1 < d.W;
        const bool wide = (d.W & 1) == 0 && in0 && in1;  // both columns inside and the pair 8-byte aligned in every row
        float2 val[NLD];
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP, y = y0 - 4 + r;
            val[t] = make_float2(-10.0f, -10.0f);
            if (live && r < ROWS && y >= 0 && y < d.H) {
                const uint32_t q = (uint32_t)(y * d.W + x);
                if (wide) {
                    val[t] = map_ld(reinterpret_cast<const float2 *>(S), q >> 1);
                } else {
                    if (in0) val[t].x = map_ld(S, q);
                    if (in1) val[t].y = map_ld(S, q + 1u);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP;
            if (live && r < ROWS) *reinterpret_cast<float2 *>(&sD[r][2 * cp]) = make_float2(val[t].x < 0 ? -10.0f : val[t].x, val[t].y < 0 ? -10.0f : val[t].y);
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int task = threadIdx.x; task < ROWS * (AM_TW / AM_NC); task += 256) {  // horizontal pass (:1402-1441): centres x0 + NC g .. of tile row r
        const int r = task / (AM_TW / AM_NC), g = task % (AM_TW / AM_NC);
        const int y = y0 - 4 + r;
        const float *q = &sD[r][AM_NC * g];  // columns x - 4 .. of the run's first centre x
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) v[i] = q[i];
        float res[AM_NC];
        bool ok[AM_NC];
        amean_run<AM_NC>(v, res, ok);
        const bool yin = y >= 3 && y < d.H - 3;
#pragma unroll
        for (int c = 0; c < AM_NC; c++) {
            const int x = x0 + AM_NC * g + c;
            const float self = v[c + 4];
            sT[r][AM_NC * g + c] = (yin && x >= 4 && x <= d.W - 4 && ok[c]) ? res[c] : (self < 0 ? -10.0f : 0.0f);  // D_tmp: -10 where invalid (:1313-1318), canonical 0 elsewhere (:1308)
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int task = threadIdx.x; task < AM_TW * (AM_TH / AM_NC); task += 256) {  // vertical pass (:1445-1484): centres y0 + NC g .. of tile column cx
        const int cx = task & (AM_TW - 1), g = task / AM_TW;
        const int x = x0 + cx;
        if (x >= d.W) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; i++) v[i] = sT[AM_NC * g + i][cx];  // rows y - 4 .. of the run's first centre y
        float res[AM_NC];
        bool ok[AM_NC];
        amean_run<AM_NC>(v, res, ok);
        const bool xin = x >= 3 && x < d.W - 3;
#pragma unroll
        for (int c = 0; c < AM_NC; c++) {
            const int y = y0 + AM_NC * g + c;
            if (y >= d.H) break;
            // untouched unless the filter produces a value.  The tile copy stands for the map's own value: it differs from it only where the
            // map holds a negative value other than -10, and every stage in front of this one (L/R check, speckle, gap) writes -10 for invalid
            float val = sD[AM_NC * g + c + 4][cx + 4];
            if (xin && y >= 4 && y <= d.H - 4 && ok[c]) val = res[c];
            map_st(dst + off, (uint32_t)(y * d.W + x), val);
        }
    }
}

// Half-resolution variant (elas.cpp:1332-1397): 4-pixel window p-2..p+1 around the centre p, ring slot = pixel index mod 4,
// plain left-to-right sums of the four slots.
__device__ __forceinline__ bool amean4(const float xs[4], float xc, float &out) {
    float w[4], f[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float t = 4.0f - absq(xs[j] - xc);
        w[j] = __builtin_fmaxf(t, 0.0f);
        f[j] = xs[j] * w[j];
    }
    const float weight_sum = w[0] + w[1] + w[2] + w[3];  // elas.cpp:1352-1353
    const float factor_sum = f[0] + f[1] + f[2] + f[3];
    if (weight_sum > 0) {
        const float dd = amean_div(factor_sum, weight_sum);
        if (dd >= 0) {
            out = dd;
            return true;
        }
    }
    return false;
}

__global__ __launch_bounds__(256) void k_amean_sub(KParams k, int nproc, const int32_t *__restrict__ blob, const float *__restrict__ src, float *__restrict__ dst) {
    const Dims &d = k.d;  // map dimensions
    const int m = blockIdx.z;
    if (blob[(m / nproc) * META_WORDS] < 3) return;
    const size_t off = map_offset(d, m, nproc);
    const float *S = src + off;
    const int x0 = blockIdx.x * PF_TW, y0 = blockIdx.y * PF_TH;
    __shared__ float sD[PF_TH + 3][PF_TW + 4];  // D_copy: rows y0-2.., columns x0-2..; invalid -> -10
    __shared__ float sT[PF_TH + 3][PF_TW];      // D_tmp after the horizontal pass: rows y0-2.., columns x0..
    for (int i = threadIdx.x; i < (PF_TH + 3) * (PF_TW + 4); i += 256) {
        const int r = i / (PF_TW + 4), c = i - r * (PF_TW + 4);
        const int y = y0 - 2 + r, x = x0 - 2 + c;
        float val = -10.0f;
        if (y >= 0 && y < d.H && x >= 0 && x < d.W) {
            val = S[(size_t)y * d.W + x];
            if (val < 0) val = -10.0f;
        }
        sD[r][c] = val;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (PF_TH + 3) * PF_TW; i += 256) {  // horizontal pass (:1335-1364): centre x = u-1, u in [3, W)
        const int r = i / PF_TW, cx = i - r * PF_TW;
        const int y = y0 - 2 + r, x = x0 + cx;
        const float self = sD[r][cx + 2];
        float out = self < 0 ? -10.0f : 0.0f;  // D_tmp: -10 where invalid, canonical 0 elsewhere
        if (y >= 3 && y < d.H - 3 && x >= 2 && x <= d.W - 2) {
            const int first = x - 2;  // window x-2..x+1 = tile columns cx..cx+3; ring slot of pixel p is p & 3
            float xs[4];
#pragma unroll
            for (int j = 0; j < 4; j++) xs[j] = sD[r][cx + ((j - first) & 3)];
            float res;
            if (amean4(xs, self, res)) out = res;
        }
        sT[r][cx] = out;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PF_TH * PF_TW; i += 256) {  // vertical pass (:1367-1396): centre y = v-1, v in [3, H)
        const int ry = i / PF_TW, cx = i - ry * PF_TW;
        const int y = y0 + ry, x = x0 + cx;
        if (y >= d.H || x >= d.W) continue;
        float val = S[(size_t)y * d.W + x];  // untouched unless the filter produces a value
        if (x >= 3 && x < d.W - 3 && y >= 2 && y <= d.H - 2) {
            const int first = y - 2;  // window rows y-2..y+1 = tile rows ry..ry+3
            float xs[4];
#pragma unroll
            for (int j = 0; j < 4; j++) xs[j] = sT[ry + ((j - first) & 3)][cx];
            float res;
            if (amean4(xs, sT[ry + 2][cx], res)) val = res;
        }
        dst[off + (size_t)y * d.W + x] = val;
    }
}

void launch_amean(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st, const float *src, float *dst) {
    if (k.d.sub)
        SV_LAUNCH(K_AMEAN, k_amean_sub, dim3((k.d.W + PF_TW - 1) / PF_TW, (k.d.H + PF_TH - 1) / PF_TH, n * nproc), dim3(256), 0, st, k, nproc, s.blob, src, dst);
    else
        SV_LAUNCH(K_AMEAN, k_amean, dim3((k.d.W + AM_TW - 1) / AM_TW, (k.d.H + AM_TH - 1) / AM_TH, n * nproc), dim3(256), 0, st, k, nproc, s.blob, src, dst);
}

// ------------------------------------------------------------------------------------------------------------
// K10 separable 7-tap median      reference: elas.cpp:1496-1560
// ------------------------------------------------------------------------------------------------------------
// Two neighbouring windows a0..a6 and a1..a7 share six values.  With s3 <= s4 the two middle values of those six, the median of
// seven is med3(seventh, s3, s4) (the 4th smallest of the seven is s3 if the seventh is below it, s4 if above, else the seventh
// itself).  The six are sorted as two triples (min3/med3/max3), and the k-th smallest of two sorted runs is
// min over i+j=k of max(A_i, B_j).  The median of a multiset does not depend on how it is found (no NaNs in a disparity map).
__device__ __forceinline__ void median7_pair(const float a[8], float &m0, float &m1) {
    const float a1 = __builtin_fminf(__builtin_fminf(a[1], a[2]), a[3]), a3 = __builtin_fmaxf(__builtin_fmaxf(a[1], a[2]), a[3]);
    const float a2 = __builtin_amdgcn_fmed3f(a[1], a[2], a[3]);
    const float b1 = __builtin_fminf(__builtin_fminf(a[4], a[5]), a[6]), b3 = __builtin_fmaxf(__builtin_fmaxf(a[4], a[5]), a[6]);
    const float b2 = __builtin_amdgcn_fmed3f(a[4], a[5], a[6]);
    const float s3 = __builtin_fminf(__builtin_fminf(__builtin_fminf(a3, b3), __builtin_fmaxf(a2, b1)), __builtin_fmaxf(a1, b2));
    const float s4 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a3, b1), __builtin_fmaxf(a2, b2)), __builtin_fmaxf(a1, b3));
    m0 = __builtin_amdgcn_fmed3f(a[0], s3, s4);
    m1 = __builtin_amdgcn_fmed3f(a[7], s3, s4);
}

// Horizontal then vertical 7-tap median through LDS tiles, out of place (src -> dst, and straight into the caller's map).
__global__ __launch_bounds__(256) void k_median(KParams k, int nproc, const int32_t *__restrict__ blob, const float *__restrict__ src, float *__restrict__ dst,
                                                float *__restrict__ user_d1, float *__restrict__ user_d2) {
    const Dims &d = k.d;
    const int m = blockIdx.z;
    const int pair = m / nproc, side = m - pair * nproc;
    if (blob[pair * META_WORDS] < 3) return;
    float *user = side == 0 ? user_d1 : user_d2;  // the last stage writes the caller's map directly; dst (the engine's own copy) only for debug dumps
    if (!user && !dst) return;
    const size_t off = map_offset(d, m, nproc);
    const float *S = src + off;
    const int x0 = blockIdx.x * PF_TW, y0 = blockIdx.y * PF_TH;
    constexpr int XO = 4;  // the tile starts at the even column x0 - 4: a thread stages one column pair (float2 loads, the column tests once)
    __shared__ __attribute__((aligned(16))) float sD[PF_TH + 6][PF_TW + 8];  // rows y0-3.., columns x0-4..
    __shared__ float sT[PF_TH + 6][PF_TW];      // D_temp after the horizontal pass (:1515-1534): rows y0-3.., columns x0..
    {  // all of a thread's tile loads are requested before the first one is used
        constexpr int ROWS = PF_TH + 6, PAIRS = (PF_TW + 8) / 2, RPP = 256 / PAIRS, NLD = (ROWS + RPP - 1) / RPP;  // 36 pairs, 7 rows per pass, 6 passes
        const int cp = threadIdx.x % PAIRS, rp = threadIdx.x / PAIRS;
        const int x = x0 - XO + 2 * cp;
        const bool live = rp < RPP, in0 = x >= 0 && x < d.W, in1 = x + 1 >= 0 && x + 1 < d.W;
        const bool wide = (d.W & 1) == 0 && in0 && in1;
        float2 val[NLD];
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP, y = y0 - 3 + r;
            val[t] = make_float2(0.0f, 0.0f);
            if (live && r < ROWS && y >= 0 && y < d.H) {
                const uint32_t q = (uint32_t)(y * d.W + x);
                if (wide) {
                    val[t] = map_ld(reinterpret_cast<const float2 *>(S), q >> 1);
                } else {
                    if (in0) val[t].x = map_ld(S, q);
                    if (in1) val[t].y = map_ld(S, q + 1u);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NLD; t++) {
            const int r = rp + t * RPP;
            if (live && r < ROWS) *reinterpret_cast<float2 *>(&sD[r][2 * cp]) = val[t];
        }
    }
    __syncthreads();
    static_assert((PF_TW & 1) == 0 && (PF_TH & 1) == 0, "two outputs per thread");
    {  // horizontal pass (:1515-1534): a thread makes columns 2c and 2c+1 of rows w, w+8, ...
        const int c2 = (threadIdx.x & 31) * 2, w = threadIdx.x >> 5;
        const int x = x0 + c2;
        const bool in0 = x >= 3 && x < d.W - 3, in1 = x + 1 >= 3 && x + 1 < d.W - 3;
#pragma unroll
        for (int t = 0; t < (PF_TH + 6 + 7) / 8; t++) {
            const int r = w + 8 * t;
            if (r >= PF_TH + 6) break;
            const int y = y0 - 3 + r;
            float a[8];
            {  // columns x - 3 .. x + 4 = tile columns c2 + 1 .. c2 + 8: five aligned pairs, the outer halves of the first and the last unused
                const float2 p0 = *reinterpret_cast<const float2 *>(&sD[r][c2]), p4 = *reinterpret_cast<const float2 *>(&sD[r][c2 + 8]);
                a[0] = p0.y, a[7] = p4.x;
#pragma unroll
                for (int j = 1; j < 4; j++) {
                    const float2 v = *reinterpret_cast<const float2 *>(&sD[r][c2 + 2 * j]);
                    a[2 * j - 1] = v.x;
                    a[2 * j] = v.y;
                }
            }
            float m0, m1;
            median7_pair(a, m0, m1);
            const bool yin = y >= 3 && y < d.H - 3;
            float2 out;  // calloc'd D_temp (:1506) outside the filtered area; invalid centres keep their value
            out.x = (yin && in0) ? (a[3] >= 0 ? m0 : a[3]) : 0.0f;
            out.y = (yin && in1) ? (a[4] >= 0 ? m1 : a[4]) : 0.0f;
            *reinterpret_cast<float2 *>(&sT[r][c2]) = out;
        }
    }
    __syncthreads();
    {  // vertical pass (:1537-1556): a thread makes rows 2p and 2p+1 of one column, p = w, w+4, ...
        const int cx = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int x = x0 + cx;
        const bool xin = x >= 3 && x < d.W - 3;
#pragma unroll
        for (int t = 0; t < PF_TH / 8; t++) {
            const int ry = 2 * (w + 4 * t);
            const int y = y0 + ry;
            if (y >= d.H || x >= d.W) continue;
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = sT[ry + j][cx];
            float m0, m1;
            median7_pair(a, m0, m1);
            float v0 = sD[ry + 3][cx + XO], v1 = sD[ry + 4][cx + XO];
            if (xin && y >= 3 && y < d.H - 3 && v0 >= 0) v0 = m0;
            if (xin && y + 1 >= 3 && y + 1 < d.H - 3 && v1 >= 0) v1 = m1;
            const uint32_t q = (uint32_t)(y * d.W + x);
            if (dst) map_st(dst + off, q, v0);
            if (user) map_st(user + (size_t)pair * d.N, q, v0);
            if (y + 1 < d.H) {
                if (dst) map_st(dst + off, q + (uint32_t)d.W, v1);
                if (user) map_st(user + (size_t)pair * d.N, q + (uint32_t)d.W, v1);
            }
        }
    }
}

void launch_median(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st, const float *src, float *dst, float *user_d1, float *user_d2) {
    dim3 grid((k.d.W + PF_TW - 1) / PF_TW, (k.d.H + PF_TH - 1) / PF_TH, n * nproc);
    SV_LAUNCH(K_MEDIAN, k_median, grid, dim3(256), 0, st, k, nproc, s.blob, src, dst, user_d1, user_d2);
}

// ------------------------------------------------------------------------------------------------------------
// final copy into the caller's maps (pairs with < 3 support points are left untouched, as elas.cpp:63-69 does)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_output(KParams k, const int32_t *__restrict__ blob, const float *__restrict__ disp, float *__restrict__ d1, float *__restrict__ d2) {  // disp: current maps
    const Dims &d = k.d;
    const int pair = blockIdx.y;
    if (blob[pair * META_WORDS] < 3) return;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= d.N) return;
    const float *s1 = disp + (size_t)(pair * 2) * d.N, *s2 = s1 + d.N;
    if (i + 3 < d.N && (d.N & 3) == 0) {
        *reinterpret_cast<float4 *>(d1 + (size_t)pair * d.N + i) = *reinterpret_cast<const float4 *>(s1 + i);
        if (d2) *reinterpret_cast<float4 *>(d2 + (size_t)pair * d.N + i) = *reinterpret_cast<const float4 *>(s2 + i);
    } else {
        for (int j = i; j < min(i + 4, d.N); j++) {
            d1[(size_t)pair * d.N + j] = s1[j];
            if (d2) d2[(size_t)pair * d.N + j] = s2[j];
        }
    }
}

void launch_output(const KParams &k, const SlotDev &s, int n, const float *src, float *d1, float *d2, hipStream_t st) {
    SV_LAUNCH(K_OUTPUT, k_output, dim3((k.d.N / 4 + 256) / 256, n), dim3(256), 0, st, k, s.blob, src, d1, d2);
}

}  // namespace sv
