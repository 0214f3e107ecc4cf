/*
 * libstereo_vision_hip.so — C ABI of the MI355X-native stereo disparity engine.
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  Two groups of entry points:
 *
 *  (A) The reference's own exported symbols, kept signature-for-signature so that the reference's
 *      ctypes binding (stereo_vision/sv.py:164-192) drives this library unchanged:
 *        generatePointCloud   reference: src/serial_includes/main/stereo_vision.cpp:565-623
 *        clean                reference: src/serial_includes/main/stereo_vision.cpp:105-114
 *        getColor             reference: src/serial_includes/main/stereo_vision.cpp:625-627
 *
 *  (B) The operator seam the reference's driver calls once per frame,
 *        Elas::process(I1, I2, D1, D2, dims)   reference: src/serial_includes/elas/elas.h:162,
 *                                               called from stereo_vision.cpp:296-318
 *      exposed as a handle-based, batched API (sv_*): the reference keeps all state in file-scope
 *      globals and function statics (stereo_vision.cpp:50-89, 307-314, 582) and processes one pair
 *      per call; here state lives in an sv_handle and one call takes B independent pairs.
 *
 * All sv_* functions return SV_OK (0) or a negative sv_status; sv_last_error() gives the text.
 * Nothing in this library calls exit().
 */
#ifndef STEREO_VISION_HIP_H
#define STEREO_VISION_HIP_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (B) batched Elas::process ------------------------------------------------------------------------ */

/* Field-for-field mirror of Elas::parameters (reference: src/serial_includes/elas/elas.h:60-145);
 * every bool is an int32 so the block is 23 four-byte words. */
typedef struct sv_params {
    int32_t disp_min;              /* elas.h:61  first disparity of the support matching's search (elas.cpp:318; negative = 0; both presets use 0) */
    int32_t disp_max;              /* elas.h:62  D = disp_max + 1, 10 <= disp_max <= 1023 */
    float support_threshold;       /* elas.h:63 */
    int32_t support_texture;       /* elas.h:64 */
    int32_t candidate_stepsize;    /* elas.h:65 */
    int32_t incon_window_size;     /* elas.h:66 */
    int32_t incon_threshold;       /* elas.h:67 */
    int32_t incon_min_support;     /* elas.h:68 */
    int32_t add_corners;           /* elas.h:69 */
    int32_t grid_size;             /* elas.h:70 */
    float beta;                    /* elas.h:71 */
    float gamma;                   /* elas.h:72 */
    float sigma;                   /* elas.h:73 */
    float sradius;                 /* elas.h:74 */
    int32_t match_texture;         /* elas.h:75 */
    int32_t lr_threshold;          /* elas.h:76 */
    float speckle_sim_threshold;   /* elas.h:77 */
    int32_t speckle_size;          /* elas.h:78 */
    int32_t ipol_gap_width;        /* elas.h:79 */
    int32_t filter_median;         /* elas.h:80 */
    int32_t filter_adaptive_mean;  /* elas.h:81 */
    int32_t postprocess_only_left; /* elas.h:82 */
    int32_t subsampling;           /* elas.h:83  (bool) half-resolution mode: disparity maps are (width/2) x (height/2) */
} sv_params;

enum sv_setting { SV_ROBOTICS = 0, SV_MIDDLEBURY = 1, SV_DRIVER = 2 };

/* SV_ROBOTICS / SV_MIDDLEBURY: the two presets of elas.h:92-143.
 * SV_DRIVER: what the reference driver actually runs (stereo_vision.cpp:307-311):
 *            MIDDLEBURY + postprocess_only_left + filter_adaptive_mean. */
void sv_params_init(sv_params *p, int setting);

typedef enum sv_status {
    SV_OK = 0,
    SV_ERR_ARG = -1,         /* bad argument / unsupported parameter value */
    SV_ERR_HIP = -2,         /* a HIP runtime call failed */
    SV_ERR_NO_DEVICE = -3,   /* no usable GPU */
    SV_ERR_UNSUPPORTED = -4, /* a test hook's input beyond what its kernel takes */
    SV_ERR_STATE = -5
} sv_status;

typedef struct sv_handle sv_handle;

/* Engine configuration beyond the ELAS parameters.  Every field is an int32; 0 always means "the default" (zero-initialise the struct and
 * set what you need).  The policy fields below n_slots used to be environment variables; the variables are still read - in ONE place,
 * engine.cpp: apply_env_overrides - and override the fields, so that a deployment can be steered without a rebuild; two handles of
 * one process can be configured differently through the fields. */
typedef struct sv_config {
    int32_t width;      /* image width  (>= 32) */
    int32_t height;     /* image height (>= 32) */
    int32_t device;     /* HIP device ordinal */
    int32_t n_workers;  /* host pool threads for the CPU stage between the two GPU phases (0 = default: the cgroup CPU quota / the
                         * affinity mask, shared between the ranks of a node, at most 16 without a quota) */
    int32_t chunk;      /* pairs per GPU launch = pairs per pipeline slot (0 = default 64, less for large images) */
    int32_t keep_debug; /* != 0: keep per-stage intermediates of the LAST processed pair for sv_debug_get */
    int32_t n_streams;  /* HIP streams the second GPU phase alternates over (0 = default 5; 4 for images of 2 M pixels and more); phase 1 has its own streams */
    int32_t n_slots;    /* buffer slots (chunks in flight) of the 3-stage pipeline (0 = default 8, within a quarter of the free HBM / 64 GB) */
    /* ---- policy (0 = automatic) */
    int32_t gpu_lattice_filter;    /* support-lattice filters: 0 auto (GPU for chunk >= 4), 1 GPU, 2 host pool             [SV_GPU_FILTER=1 / SV_HOST_FILTER=1] */
    int32_t gpu_triangulation;     /* who triangulates: 0 auto (pool and GPU balanced by the pool's backlog; GPU alone with < 3 host threads),
                                      1 GPU, 2 host pool, 3 a fixed share of gpu_triangulation_pct percent on the GPU, 4 balanced by the pool's
                                      backlog whatever the number of host threads                                        [SV_GPU_DELAUNAY=1/0, SV_GPU_DELAUNAY_PCT=n, SV_GPU_DELAUNAY_AUTO=0] */
    int32_t gpu_triangulation_pct; /* the share for mode 3 (1..100) */
    int32_t resident;              /* the GPU's share without the support lists ever leaving the device: 0 auto (on), 2 off  [SV_RESIDENT=0] */
    int32_t dg_sub_max;            /* > 0: vertices a set may have to be triangulated whole in LDS (default 4000; experiments, tests) [SV_DG_SUBMAX] */
    int32_t dg_max_points;         /* > 0: largest vertex set the GPU kernels take, larger ones go to the pool (tests)       [SV_GPU_DELAUNAY_MAX] */
    int32_t affinity;              /* host threads on the CPUs of the GPU's NUMA node: 0 auto (when the node has enough allowed CPUs), 2 never [SV_NO_AFFINITY=1] */
    int32_t inline_latency_path;   /* single pairs on a chunk-1 handle driven by the calling thread: 0 auto (on), 2 off       [SV_NO_INLINE=1] */
    int32_t event_sync;            /* how host threads wait for the GPU: 0 auto (3 for chunk >= 4, 2 below), 1 hipEventBlockingSync, 2 spin, 3 ask the event + 40 us naps [SV_EVENT_SYNC=block|spin|poll] */
    int32_t share_sliced;          /* != 0: a balanced GPU share as a slice of every chunk instead of whole chunks (round-2 behaviour, non-resident only) [SV_GPU_DELAUNAY_SLICED=1] */
    int32_t latency_split;         /* single pairs: 0 automatic - each triangulation in quarters (seven helper cores) or halves (four) on pool threads pinned to
                                      cores that share the calling thread's L3 cache, when the host has such cores within the process's mask (and affinity != 2),
                                      otherwise as 3; 1 halves / 2 quarters of the top-level cuts on pool threads in any case; 3 each triangulation on one thread
                                      [SV_LATENCY_SPLIT=0..3] */
    int32_t host_copies;           /* host-memory batches (sv_submit_batch_host...): who moves images and maps over PCIe.  0 auto = 2 where the runtime
                                      allows it, 1 hipMemcpyAsync (the runtime picks an SDMA engine per copy - the directions can end up sharing
                                      one), 2 engine-addressed copies (csrc/dma_lanes.cpp): uploads and downloads on SDMA engines of their own,
                                      queued ahead of the kernels                                                          [SV_HOST_COPIES=1|2] */
    int32_t reserved[4];           /* must be 0 (sv_create checks) */
} sv_config;

int sv_create(const sv_params *params, const sv_config *cfg, sv_handle **out);
int sv_destroy(sv_handle *h);
/* What the handle decided at creation (defaults depend on the image size, the free memory and the CPU quota). */
enum sv_query_key {
    SV_Q_HOST_THREADS = 0,       /* size of the host pool */
    SV_Q_CHUNK = 1,              /* pairs per GPU launch */
    SV_Q_SLOTS = 2,              /* chunks in flight */
    SV_Q_GPU_LATTICE_FILTER = 3, /* 1: support-lattice filters on the GPU, 0: on the host pool */
    SV_Q_GPU_TRIANGULATION = 4,  /* 1: Delaunay divide-and-conquer on the GPU (few host threads), 0: on the host pool */
    SV_Q_GPU_TRIANGULATION_FALLBACKS = 6, /* vertex sets handed to the GPU kernel's share that the host triangulated after all (coincident points,
                                             whose survivor the reference's quicksort decides; more vertices than the kernels take) */
    SV_Q_NUMA_BOUND = 7,         /* 1: the handle's host threads are bound to the CPUs of the GPU's NUMA node (SV_NO_AFFINITY=1 disables) */
    SV_Q_RESIDENT = 8,           /* 1: the GPU's share of the chunks is built "resident" - the support lists never leave the device: sort, duplicate scan,
                                    k-d order and triangulation in one kernel after the lattice filter; the host only reads 8 meta words per pair */
    SV_Q_HOST_COPIES = 9,        /* who moves host-memory batches over PCIe: 0 not decided yet (no such batch so far), 1 hipMemcpyAsync,
                                    2 engine-addressed SDMA copies (sv_config.host_copies) */
    SV_Q_LATENCY_SPLIT = 10,     /* single pairs: 0 each triangulation on one thread, 1 in halves, 2 in quarters (sv_config.latency_split as resolved at creation) */
    SV_Q_GPU_TRIANGULATION_SHARE = 5 /* per mille of the pairs so far whose triangulations the GPU kernel built (in the host mode the
                                        dispatcher hands it a share of a chunk while the pool is behind; results are identical) */
};
int sv_query(const sv_handle *h, int what);
const char *sv_last_error(const sv_handle *h); /* h may be NULL: error of the last failed sv_create */

/* B independent pairs, images and maps in DEVICE memory (HBM):
 *   left/right : uint8  [B][height][stride]   rectified gray rows (what Elas::process receives as I1/I2)
 *   d1 / d2    : float  [B][height][width]    disparity maps (what Elas::process writes to D1/D2); with params.subsampling
 *                                             the maps are [B][height/2][width/2] (elas.h:160-161);
 *                                             d2 may be NULL.  Invalid pixels are -10 (elas.cpp:823-824, 987-991).
 *   status     : int32  [B] host array, may be NULL; per pair: number of support points, or <3 when the
 *                reference would have printed "ERROR: Need at least 3 support points!" (elas.cpp:63-69) — the
 *                maps of such a pair are left untouched, as the reference leaves them.
 * The call returns when all B pairs are complete (outputs visible to every stream of the device). */
int sv_process_batch_device(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status);

/* Streaming form: sv_submit_batch_device enqueues a batch and returns at once; batches are processed in submission order and
 * flow through the same pipeline back to back (the first chunks of batch k+1 overlap the last chunks of batch k).  sv_wait
 * returns when every submitted batch is complete.  Buffers must stay valid until then.  sv_process_batch_device == submit + wait. */
int sv_submit_batch_device(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status);
int sv_wait(sv_handle *h);
/* Returns when the n oldest batches submitted since the last sv_wait are complete; later ones keep running (batches complete in
 * submission order).  For consumers that take finished batches while the engine computes the next - e.g. a chunked gather of the
 * maps on one rank that overlaps the other chunks' kernels.  Not to be called concurrently with sv_wait. */
int sv_wait_batches(sv_handle *h, int n);

/* Same call with HOST memory in and out - the form of the reference's seam, which takes host pointers (elas.h:162, call site
 * stereo_vision.cpp:313).  The batch streams through the pipeline chunk by chunk: the images of chunk k+1 go up and the maps of
 * chunk k-1 come down on copy streams while chunk k computes; nothing is allocated per call.  Page-locked ("pinned") caller
 * memory - sv_host_alloc, hipHostMalloc, hipHostRegister, torch pin_memory() - is the DMA source / target itself; pageable
 * memory is detected and goes through page-locked staging buffers of the handle (one extra host copy each way).  Maps of pairs
 * with < 3 support points are not written (the reference leaves them untouched, elas.cpp:63-69).
 * sv_submit_batch_host is the streaming form (buffers stay valid and untouched until sv_wait). */
int sv_process_batch_host(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status);
int sv_submit_batch_host(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status);
/* The same with the DRIVER's output format: dmap = saturate(round_half_even(4 * D1)) as uint8 [B][height][width] (what the reference's
 * generateDisparityMap returns: leftdpf.convertTo(dmap, CV_8UC1, 4.0), stereo_vision.cpp:316; [B][height/2][width/2] with
 * params.subsampling), converted on the device: a quarter of the bytes come back over PCIe.  Images of pairs with < 3 support
 * points are not written.  Page-locked or pageable memory, like above. */
int sv_process_batch_host_dmap(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, uint8_t *dmap, int32_t *status);
int sv_submit_batch_host_dmap(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, uint8_t *dmap, int32_t *status);
/* Page-locked host memory for the calls above (NULL on failure). */
void *sv_host_alloc(size_t bytes);
void sv_host_free(void *p);

/* Single pair with the exact argument meaning of Elas::process (elas.h:153-162): dims = {width, height, bytes per line};
 * host pointers, like the reference.  On a chunk = 1 handle the calling thread drives the pair itself through persistent
 * device buffers (latency mode). */
int sv_elas_process(sv_handle *h, const uint8_t *I1, const uint8_t *I2, float *D1, float *D2, const int32_t *dims);

/* Test hooks on a live handle (waits for submitted work first).  Keys: "ccl_cap" (runs a band of the speckle stage may hold before a
 * map takes the per-pixel path), "rt_cap" (triangles a raster tile list may hold), "host_force_staging" (host-memory jobs take the
 * pageable route whatever the caller's memory is), "ns_bound" (vertices the next resident launches request LDS for: smaller sets than
 * the chunk has are handed to the host stage), "pool_sleep" (latency handles: pool threads sleep instead of polling), "lat_trace"
 * (wall-clock split of the latency path, printed by sv_destroy), "dma_selftest_fail" (the DMA lanes' self-test reports a failure: the
 * runtime's copies take over); single pairs: "latency_pin" (0: the polling pool threads stay where the pool runs) [SV_LATENCY_PIN],
 * "lat_runtime_copies" (lattice / blob copies through hipMemcpyAsync instead of the copy kernel) [SV_LAT_RUNTIME_COPIES],
 * "lat_filter_alone" (the lattice filters on the calling thread alone) [SV_LAT_FILTER_ALONE]; SV_LAT_WAKE_LEAD_US: how long before a
 * frame that is due the sleeping helpers poll again (default: period / 16 within 0.3 - 2 ms).  Returns SV_OK or SV_ERR_ARG for an unknown key. */
int sv_debug_set(sv_handle *h, const char *key, int value);

/* Per-stage intermediates of the last pair processed (cfg.keep_debug != 0).  Names and layouts follow
 * oracle/elas_oracle.h: desc1 desc2 dcan_raw support tri1 tri2 planes1 planes2 grid1 grid2 wta1 wta2 lr1 lr2
 * speckle1 gap1 amean1 final1 ...  Returns the byte count, -1 unknown name, -2 cap too small. */
long sv_debug_size(sv_handle *h, const char *name);
long sv_debug_get(sv_handle *h, const char *name, void *out, long cap);

/* Work counters of the two matching kernels (separate instantiations of the kernels; off by default).  mode 1: enable and
 * reset, 0: disable, -1: leave as is; out (may be NULL) receives uint64[8] = {dense candidates evaluated (16-byte SADs),
 * dense pixels matched, support energies evaluated (64-byte SADs), pixels whose plane band took the straight-line path /
 * the scalar-bounded path / the per-lane path of dense_match, trips of its grid-candidate loop per wavefront / per lane (two
 * candidates per trip: lane trips / (64 x wavefront trips) = lane utilisation of that loop)} since the last reset.  Waits for submitted work. */
int sv_debug_counters(sv_handle *h, int mode, uint64_t *out);

/* Per-kernel device timings (HIP events on the worker streams) accumulated since the last reset.
 * names/ms/calls are parallel arrays written up to cap entries; returns the number of kernels known. */
int sv_kernel_times(sv_handle *h, const char **names, double *total_ms, int64_t *calls, int cap);
void sv_kernel_times_reset(sv_handle *h);
void sv_kernel_timing_enable(sv_handle *h, int on);
/* Restricts the timing to the named kernels ("dense_match,support_match", names as sv_kernel_times reports them; NULL or ""
 * = all): every timed launch costs two event records on its stream (all kernels timed: 1-2 % of the throughput). */
int sv_kernel_timing_select(sv_handle *h, const char *names);

/* Host-side stages exposed for tests (they run on the CPU in the product as well, between the two GPU phases):
 * the in-place support-point filters + corner points (elas.cpp:152-264, 413-433) and the Delaunay
 * triangulation (elas.cpp:442-501 -> Triangle "zQB"). */
int sv_host_support_filter(const sv_params *p, int16_t *dcan, int width, int height, int32_t *support, int cap);
/* The same filters with the lattice shared between `threads` threads, as single-pair calls do with the pool threads that sit next to the
 * calling thread (csrc/host_stage.cpp: the order-dependent filter splits into a parallel classification and a short serial pass). */
int sv_host_support_filter_threads(const sv_params *p, int16_t *dcan, int width, int height, int32_t *support, int cap, int threads);
int sv_host_delaunay(const int32_t *xy, int n, int32_t *tri_out, int cap);
/* Test hook: exhaustive comparison, on the current device, of the adaptive-mean kernel's division shortcut (v_rcp_f32 + one FMA
 * correction; kernels.hip: amean_div) with the IEEE division: every float mantissa, both signs, 31 exponents, the sixteen divisors
 * a weight sum can be.  Returns the number of differing quotients (0 = the shortcut is exact), < 0 on a HIP error;
 * *control = the same count for a * rcp(d) without the correction (non-zero: the comparison can fail). */
long long sv_debug_check_amean_div(unsigned int *first_a_bits, unsigned int *first_d_bits, long long *control);
/* Same triangulation with the two halves of the top-level cut built by two threads, as the engine does in latency mode
 * (chunk = 1); helper_delay_us > 0 delays the helper thread so that the caller ends up doing both halves itself. */
int sv_host_delaunay_split(const int32_t *xy, int n, int32_t *tri_out, int cap, int helper_delay_us);
/* Same with `depth` levels of the recursion shared (2: four quarters on four threads, what a latency handle with >= 7 pool
 * threads does). */
int sv_host_delaunay_par(const int32_t *xy, int n, int32_t *tri_out, int cap, int depth, int helper_delay_us);
/* Test hook: the divide-and-conquer phase of that triangulation on the GPU (csrc/delaunay_gpu.hip; sort, duplicate scan and k-d
 * ordering on the host), `reps` copies of the set in one launch, kernel time in *kernel_ms (may be NULL).  n <= 4000: one workgroup per set, mesh in LDS;
 * larger sets (<= 256 000): subtrees in LDS, upper merges in a global-memory mesh (SV_DG_SUBMAX lowers the 4000 for tests). */
int sv_gpu_delaunay(const int32_t *xy, int n, int32_t *tri_out, int cap, int reps, double *kernel_ms);

/* Test hook (no GPU needed): the size of the host pool a handle gets by default in this process (cgroup CPU quota or affinity mask,
 * shared between LOCAL_WORLD_SIZE ranks; a mask already as narrow as one rank's share is not divided again).  ignore_quota != 0: as
 * on a host without a CPU quota. */
int sv_default_host_threads(int ignore_quota);

/* Test hooks: the preparation of a vertex set for the triangulation - (x, y) sort, duplicate scan, k-d order (reference:
 * triangle.cpp:5183-5360, 5889-5903) - on the host and on the GPU (csrc/delaunay_gpu.hip: dg_prepare; vertices on the support lattice
 * of a width x height image with lattice step `step` and disparities <= disp_max, n <= 4096).  Both write the ids of the m surviving
 * vertices in the order the divide-and-conquer recursion consumes them and return m.  Coincident vertices: the host keeps the one the
 * reference's quicksort puts first; the GPU form keeps the lowest id when they carry the same disparity (disp[i], may be NULL = unknown)
 * - they are then the same support point twice and interchangeable - and returns -1 (a set it leaves to the host) otherwise. */
int sv_host_kd_order(const int32_t *xy, int n, int32_t *ids_out);
int sv_gpu_kd_order(const int32_t *xy, const int32_t *disp, int n, int width, int height, int step, int disp_max, int32_t *ids_out);

/* ---- (A) the reference's exported symbols ------------------------------------------------------------- */

typedef struct {
    double x, y, z;
} Double3; /* reference: src/common_includes/structs.h:14-16 */

typedef struct {
    unsigned char x, y, z, w;
} Uchar4; /* reference: src/common_includes/structs.h:18-20 */

/* reference: stereo_vision.cpp:565-623.  left/right: BGRA uint8 [height][width][4].  Returns the library-owned
 * point array [W*H] of the FIRST call's width x height (valid until the next call / clean()).  State is frozen at the
 * first call, like the reference's function-static init (stereo_vision.cpp:582); later frames of another size are resized
 * to it (cv::resize INTER_LINEAR restated, :590-591).  removeSky / subsampling are not read (see sv_legacy_set_subsampling). */
Double3 *generatePointCloud(unsigned char *left, unsigned char *right, char *CAMERA_CALIBRATION_YAML, int width, int height, bool kittiCalibration,
                            bool objectTracking, bool graphics, bool display, int scale, int pc_extrapolation, const char *YOLO_CFG,
                            const char *YOLO_WEIGHTS, const char *YOLO_CLASSES, bool removeSky, bool subsampling);
void clean(void);       /* reference: stereo_vision.cpp:105-114 (without the reference's exit(0)) */
Uchar4 *getColor(void); /* reference: stereo_vision.cpp:625-627 */

/* Last disparity image of the legacy path as the reference's `dmap` (u8 = saturate(round(4*d)), stereo_vision.cpp:316). */
const unsigned char *sv_legacy_last_dmap(int *width, int *height);
/* The 4x4 disparity-to-depth matrix Q the legacy path uses (row major), NULL before the first frame. */
const double *sv_legacy_Q(void);
/* Rectification remap of the gray images in front of the matcher: findRectificationMap's initUndistortRectifyMap maps
 * (stereo_vision.cpp:477-478) applied with cv::remap(INTER_LINEAR) - the call the reference has commented out at :341, so OFF
 * by default here as well.  Call before the first generatePointCloud (the state is frozen there, :582).  OpenCV arithmetic
 * restated (parity unpinned, like the gray conversion). */
void sv_legacy_set_rectify(int on);
/* Half-resolution mode of the legacy path (Elas::parameters::subsampling, set from the driver's `subsample`, stereo_vision.cpp:309).
 * generatePointCloud does NOT read its arguments 15 and 16 (removeSky, subsampling): the reference's own binding passes only 14
 * (stereo_vision/sv.py:180,189), so those slots are undefined under "sv.py drives it unchanged".  Call before the first frame. */
void sv_legacy_set_subsampling(int on);
/* HIP device of the legacy path (default 0).  Call before the first frame. */
void sv_legacy_set_device(int device);
/* The four maps lmapx, lmapy, rmapx, rmapy as [4][height][width] floats (host), NULL unless rectification is on. */
const float *sv_legacy_rectify_maps(void);
/* Test hook: the gray images of the last frame as the matcher received them (after the remap if it is on); [height][width] each. */
int sv_legacy_last_gray(unsigned char *left, unsigned char *right);
/* Batched disparity -> point cloud on the device, for callers of the batch API: the driver's conversion dmap = saturate(
 * round_half_even(4*d)) (stereo_vision.cpp:316) followed by publishPointCloud's reprojection pos = Q*[x y dmap 1]^T,
 * (X,Y,Z) = pos.xyz/pos.w in double (:233-256) for every pixel, and optionally the CUDA variant's robot-frame transform
 * XR*(X,Y,Z)+XT (parallel_includes/main/stereo_vision.cu:188-212).
 *   disp       : float  [B][height][width]     device (e.g. d1 of sv_process_batch_device)
 *   Q16        : double [16] HOST, row major   (e.g. sv_legacy_Q(), or your own stereoRectify result)
 *   XR9 / XT3  : double [9] / [3] HOST, row major; both NULL = no transform (the serial driver)
 *   dmap_out   : uint8  [B][height][width]     device, may be NULL
 *   points_out : double [B][height][width][3]  device
 * Runs on the device's default stream and returns when the points are complete. */
int sv_reproject_batch_device(const float *disp, int batch, int width, int height, const double *Q16, const double *XR9, const double *XT3, unsigned char *dmap_out,
                              double *points_out);
/* The driver's 8-bit disparity image alone: dmap = saturate(round_half_even(4 * d)) (leftdpf.convertTo(dmap, CV_8UC1, 4.0),
 * stereo_vision.cpp:316) for `count` floats in device memory, enqueued on `stream` (a hipStream_t, NULL = the default stream) and NOT
 * waited for - e.g. in front of a gather of finished maps, which then moves a quarter of the bytes. */
int sv_disparity_to_u8_device(const float *disp, size_t count, unsigned char *dmap_out, void *stream);
/* Mean 3-D position of the cloud inside each detector box (x, y, w, h in pixels), i.e. what publishPointCloud hands to its
 * viewer for every tracked object (stereo_vision.cpp:261-278; the boxes come from a detector the caller runs - the
 * reference's YOLO weights are not part of this library).  boxes: int32 [n][4]; out: double [n][3] = (X, Y, Z) sums over
 * columns [clamp(x), clamp(x+w)) outer and rows [clamp(y), clamp(y+h)) inner of the last frame's points, divided by the
 * box's pixel count - the reference's summation order, so the doubles are the reference's.  Returns 0, or -1 before the
 * first frame / on bad arguments. */
int sv_legacy_box_means(const int32_t *boxes, int n, double *out);
/* Test hook: Q (and P1,P2) of the stereoRectify restatement for a calibration file; K1,K2 are divided by `scale` first
 * (stereo_vision.cpp:364-376).  variant 1 = OpenCV 4.x rule set (the product), 0 = pre-3.4.2 rule set. */
int sv_debug_stereo_rectify(const char *yaml, int image_w, int image_h, double scale, int variant, double *Q16, double *P1P2_24);

/* Test hook: cv::initUndistortRectifyMap(K, D(k1,k2,p1,p2,k3), R, P(3x4), (w,h), CV_32F) as the legacy path restates it. */
int sv_debug_undistort_map(const double *K9, const double *D5, const double *R9, const double *P12, int w, int h, float *mapx, float *mapy);

#ifdef __cplusplus
}
#endif
#endif
