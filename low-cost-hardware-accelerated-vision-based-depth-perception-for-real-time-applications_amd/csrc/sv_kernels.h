// Internal interface between the host engine (engine.cpp) and the gfx950 kernels (kernels.hip).
// Not part of the C ABI (that is include/stereo_vision_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <stdexcept>
#include <string>

namespace sv {

// Geometry of one stereo pair as the kernels see it.  All maps are dense row-major [H][W].
struct Dims {
    int W, H, N;             // N = W*H
    int step, Wc, Hc;        // support lattice (elas.cpp:376-386): Wc x Hc candidates, `step` px apart
    int grid_size, gw, gh;   // candidate grid cells (elas.cpp:88-90)
    int ncell, MW;           // gw*gh cells, MW 32-bit mask words per cell (bit d set <=> disparity d is a candidate)
    int D, disp_max;         // D = disp_max + 1
    int disp_min;            // max(Elas::parameters::disp_min, 0): first disparity the support matching scans (elas.cpp:318; nothing else uses it)
    int max_pts, max_tri;    // capacities of the per-pair support / triangle arrays
    int sub;                 // half-resolution mode (Elas::parameters::subsampling, elas.h:83-85)
    int Wm, Hm, Nm;          // disparity MAP size: W/2 x H/2 when sub, else W x H.  The post-matching kernels get a KParams whose
                             // W/H/N already ARE the map size (engine: kp_map); W/H/N of the matching kernels are the image size
};

// Everything a kernel needs besides buffers; passed by value.
struct KParams {
    Dims d;
    float support_threshold;
    int support_texture;
    int lr_threshold;
    int match_texture;
    int plane_radius;
    int prior[16];           // P[delta] for delta <= plane_radius (elas.cpp:828-831), computed on the host
    float speckle_sim;
    int speckle_size;
    int gap_width;
    int add_corners;
    int ccl_cap;             // runs a 16-row band may hold in the LDS tables of k_ccl_band before its map takes the slow path
    int rt_cap;              // triangles a raster tile list may hold before its map falls back to global atomics (<= 512)
    uint32_t cell_mul;       // (u * cell_mul) >> 16 == u / grid_size for every column u < W (checked on the host), or 0: divide
};

// Scratch of the GPU triangulation for sets that do not fit LDS (delaunay_gpu.hip, launch_delaunay_gpu_large): per set 2 * cap
// triangles of 24 bytes, cap (x, y) pairs, node results (delaunay_scratch_bytes).
struct DelaunayScratch {
    void *tri = nullptr;
    int32_t *xy = nullptr;
    uint32_t *res = nullptr;
    int cap = 0;  // vertices per set the scratch holds
    void *prep = nullptr;  // resident chunks: scratch of the large sets' preparation (delaunay_prep_large_bytes per set), or nullptr
};

// Device buffers of one worker slot, each holding `cap` pairs back to back.
constexpr int META_WORDS = 8;

struct SlotDev {
    int cap;
    uint8_t *grad;      // [cap][2][2][H][P] Sobel gradient planes (image 0 = left, 1 = right; du, dv; padded rows, kernels.hip) -
                        //                   the 16-byte descriptors are assembled from them in LDS, never stored
    int16_t *dcan;      // [cap][Wc][Hc]     raw support lattice, TRANSPOSED (u major) for the host filters
    int32_t *fsup;      // [cap][max_pts][3] support points from the on-GPU lattice filter (when it is used)
    DelaunayScratch dg; // [cap][2] sets, when the handle may triangulate large sets on the GPU (cap 0: it may not)
    int32_t *fnsup;     // [cap]
    void *flt_ws;       // workspace of the on-GPU lattice filter (support_filter_ws_bytes)
    int32_t *blob;      // host-stage results of the chunk, one H2D copy: [cap][8] meta words, then tightly packed data.
                        //   meta of pair p: [0] #support points  [1] offset of its (u,v,d) triples
                        //                   [2] #triangles left  [3] offset of their corner indices
                        //                   [4] #triangles right [5] offset           (offsets in int32 units from blob)
    float4 *trirec;     // [cap][2][max_tri]  (a, b, c, valid) of the side's own plane
    void *rrec;         // [cap][2][max_tri]  raster records (k_planes -> raster kernels), 36 B each
    int32_t *tile_cnt;  // [cap][2][ntile] triangles binned per 64x32 raster tile (same allocation as gmaskA, right behind it)
    int32_t *tile_list; // [cap][2][ntile][512] their indices
    float *planes;      // [cap][2][max_tri][6] t1a t1b t1c t2a t2b t2c (kept for parity tests)
    uint32_t *gmaskA;   // [cap][2][ncell][MW] support marks
    uint32_t *gmaskB;   // [cap][2][ncell][MW] after the 3x3 flat dilation
    int32_t *tri_id;    // [cap][2][N]  last triangle covering a pixel, -1 = none; reused as CCL labels
    int16_t *wta;       // [cap][2][N]  WTA disparity (-1 / -10 invalid); 2 bytes per pixel: the values are integers < 32768
    float *disp;        // [cap][2][N]  L/R-checked maps; speckle and gap stages work in place
    float *tmp;         // [cap][2][N]  second map buffer: the fused filters ping-pong disp <-> tmp (CCL counters before that)
    int32_t *csize;     // [cap][2][N]  slow-path CCL only: run lengths (at run-start pixels); component sizes accumulate in `tmp`
    void *ccl_ws;       // run records, union-find and border masks of the speckle stage (ccl_ws_bytes)
    mutable uint32_t ccl_epoch = 0;  // launches of the speckle stage on this workspace so far: the overflow marks of one launch mean nothing to the next
    unsigned long long *counters;  // work counters of the matching kernels (CounterId), nullptr = not counting (the normal case)
};

enum CounterId { CNT_DENSE_CANDIDATES = 0, CNT_DENSE_PIXELS, CNT_SUPPORT_ENERGIES, CNT_DENSE_BAND_FULL, CNT_DENSE_BAND_PART, CNT_DENSE_BAND_SLOW, CNT_DENSE_GRID_WAVE_TRIPS, CNT_DENSE_GRID_LANE_TRIPS, CNT_COUNT = 8 };

// ---- launch wrappers (kernels.hip).  `n` = pairs in this launch; `nproc` = maps per pair to post-process (1 or 2).
void launch_sobel(const KParams &k, const uint8_t *left, const uint8_t *right, size_t in_pair_stride, int stride, const SlotDev &s, int n, hipStream_t st);
size_t grad_bytes_per_pair(const KParams &k);
int launch_check_amean_div(unsigned long long *d_mism, hipStream_t st);  // exhaustive check of the adaptive mean's division shortcut (kernels.hip: amean_div)
void launch_copy_block(void *dst, const void *src, size_t bytes, hipStream_t st);  // latency mode: small host <-> device blocks in stream order, by a kernel
void launch_expand_debug(const KParams &k, const SlotDev &s, int n, uint8_t *desc, hipStream_t st);  // debug: the descriptor images for the stage snapshot
void launch_support(const KParams &k, const SlotDev &s, int n, hipStream_t st);
size_t support_filter_ws_bytes(const KParams &k, int cap);
size_t ccl_ws_bytes(const KParams &k, int maps_cap);
size_t ccl_lds_bytes(const KParams &k);
void launch_support_filter(const KParams &k, int win, int thr, int need, const SlotDev &s, int n, hipStream_t st);
size_t grid_masks_words(const KParams &k, int cap);  // gmaskA and tile_cnt are ONE allocation of grid_clear_bytes (cleared by one memset)
size_t raster_tiles(const KParams &k);
size_t grid_clear_bytes(const KParams &k, int cap);
void launch_grid(const KParams &k, const SlotDev &s, int n, int max_points, hipStream_t st);
void launch_triangles(const KParams &k, const SlotDev &s, int n, int max_points, hipStream_t st);
void launch_dense(const KParams &k, const SlotDev &s, int n, hipStream_t st);
void launch_lr(const KParams &k, const SlotDev &s, int n, hipStream_t st, float *user_d2, bool keep_right);
void launch_speckle(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st);
void launch_gap_rows(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st);
void launch_gap_cols(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st);
void launch_amean(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st, const float *src, float *dst);
void launch_median(const KParams &k, const SlotDev &s, int n, int nproc, hipStream_t st, const float *src, float *dst, float *user_d1, float *user_d2);
void launch_output(const KParams &k, const SlotDev &s, int n, const float *src, float *d1, float *d2, hipStream_t st);
int launch_disp_to_u8(const float *disp, size_t n, unsigned char *out, hipStream_t st);  // legacy_kernels.hip: saturate(round_half_even(4 d)), stereo_vision.cpp:316

// GPU triangulation (delaunay_gpu.hip): one workgroup per vertex set, sets[s] = {first entry in order/xy, vertices after the
// duplicate scan, entries of xy, offset of the set's triangle list in tri_out (in int32 units)}
size_t delaunay_gpu_lds_bytes(int m, int npts);
int delaunay_gpu_max_points();
void launch_delaunay_blob(int32_t *blob, int n_pairs, size_t lds, int sub_max, hipStream_t st);  // sets of more than sub_max points are left to ..._large
int launch_delaunay_gpu(const int4 *sets, int nsets, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count, size_t lds, int narrow, hipStream_t st,
                        long long *d_level_clock = nullptr);  // narrow: all coordinate differences < 2^14
// The resident form (k_delaunay_resident): support lists straight from the lattice filter's device buffers, preparation (sort,
// duplicate scan, k-d order) and triangulation in one kernel, the pair's part of the blob laid out on the device at a fixed place
// (pair_words per pair behind the meta words).  Sides it reports with a triangle count of -1 (coincident points) are the host's.
size_t delaunay_resident_lds_bytes(int W, int H, int step, int disp_max, int m);
int delaunay_prep_max_points();
void launch_delaunay_resident(const int32_t *fsup, const int32_t *fnsup, int32_t *blob, int cap, int max_pts, int pair_words, int n_pairs, int ns_max, int sub_max, int W, int H, int step,
                              int disp_max, hipStream_t st, int large_min = 0x7FFFFFFF, int large_cap = 0);
// ... and its sets beyond LDS (large_min < vertices <= large_cap: left alone by the kernel above): preparation in global memory, then the cut path
size_t delaunay_prep_large_bytes(int W, int H, int step, int disp_max, int cap);
void launch_delaunay_resident_large(int32_t *blob, int n_pairs, int sub_max, int large_cap, int W, int H, int step, int disp_max, void *prep_scratch, const DelaunayScratch &scratch,
                                    hipStream_t st);
int launch_delaunay_prepare_test(const int32_t *d_xy, const int32_t *d_dsp, int n, int32_t *d_ord_out, int W, int H, int step, int disp_max, hipStream_t st);  // d_dsp: disparities, may be nullptr
// ... sets that do not fit LDS (more than delaunay_gpu_max_points() vertices): subtrees in LDS, the upper merges in a global-memory
// mesh.  Scratch per set: 2 * cap triangles of 24 bytes, cap (x, y) pairs, node results (delaunay_scratch_bytes).
int delaunay_gpu_large_max_points();
int delaunay_gpu_cut_max();  // a large set is cut into at most 2^this subtrees of at most sub_max vertices
size_t delaunay_scratch_bytes(int cap, int nsets, size_t *tri_bytes, size_t *xy_bytes, size_t *res_bytes);
int launch_delaunay_gpu_large(const int4 *sets, int nsets, const int32_t *order, const int32_t *xy, int32_t *tri_out, int32_t *tri_count, int m_max, int sub_max,
                              const DelaunayScratch &scratch, int narrow, hipStream_t st);
void launch_delaunay_blob_large(int32_t *blob, int n_pairs, int ns_max, int sub_max, const DelaunayScratch &scratch, hipStream_t st);

// names of the kernels behind each wrapper, in launch order, for timing reports
enum KernelId {
    K_DESCRIPTOR = 0, K_SUPPORT, K_SUPPORT_FILTER, K_GRID_MARK, K_GRID_DILATE, K_PLANES, K_TRIANGLES, K_DENSE, K_LR,
    K_DELAUNAY, K_CCL_BAND, K_CCL_FINISH, K_GAP_ROWS, K_GAP_COLS, K_AMEAN, K_MEDIAN, K_OUTPUT, K_COUNT
};
const char *kernel_name(int id);

// Optional per-launch timing hook: when set, each kernel launch is bracketed by the callback (before=true/false).
struct LaunchHook {
    void (*fn)(void *ctx, int kernel_id, bool before, hipStream_t st);
    void *ctx;
};
extern thread_local LaunchHook g_launch_hook;

// Dynamic LDS above the 64 KB default needs the function attribute raised once per device (the attribute belongs to the
// function's code object on that device).  Throws when the runtime refuses: launching anyway would fail silently and leave
// the previous chunk's results in the output buffers.
template <class Kernel>
inline void ensure_dynamic_lds(Kernel kernel, size_t bytes, std::atomic<size_t> (&granted)[64], const char *name) {
    if (bytes <= 64 * 1024) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t> &g = granted[dev & 63];
    if (bytes <= g.load()) return;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) throw std::runtime_error(std::string(name) + ": " + std::to_string(bytes) + " bytes of LDS refused: " + hipGetErrorString(e));
    g.store(bytes);
}

#define SV_LAUNCH(id, kernel, grid, block, shmem, st, ...)                         \
    do {                                                                           \
        if (g_launch_hook.fn) g_launch_hook.fn(g_launch_hook.ctx, id, true, st);   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, st, __VA_ARGS__);           \
        if (g_launch_hook.fn) g_launch_hook.fn(g_launch_hook.ctx, id, false, st);  \
    } while (0)

}  // namespace sv
