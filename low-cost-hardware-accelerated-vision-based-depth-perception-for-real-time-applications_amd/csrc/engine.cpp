// Batch engine + C ABI (group B of include/stereo_vision_hip.h).
//
// Execution model (MI355X-first, not the reference's one-pair-per-call globals).  One sv_handle = one GPU.  A batch is cut
// into chunks of `chunk` pairs that flow through a three-stage software pipeline over a ring of `n_slots` buffer slots:
//
//   stage 1  "issuer" thread, stream P1:   descriptors + support matching for chunk k, D2H of the small support lattices
//   stage 2  host pool (W threads):         per pair the order-dependent in-place lattice filters and the two Delaunay
//                                           triangulations (work the reference also does on the CPU), packed by atomic bump
//                                           allocation into the slot's pinned blob; a "dispatcher" thread feeds the pool as
//                                           soon as a chunk's lattices have landed
//   stage 3  "finisher" thread, streams P2: one H2D of the blob, then grid, plane fit + raster, dense matching, L/R check,
//                                           speckle, gap interpolation, adaptive mean, median, output
//
// Stage 1 runs ahead as far as the ring allows, so the pool always has work and phase-1 kernels of later chunks overlap
// phase-2 kernels of earlier ones on the GPU.  Every launch covers a whole chunk (batch index in blockIdx.z / .y).
// No allocation, hipMalloc or device-wide synchronisation on the per-batch path.
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/stereo_vision_hip.h"
#include "dma_lanes.h"
#include "host_stage.h"
#include "sv_kernels.h"

using namespace sv;

namespace {

thread_local std::string g_create_error;

struct TimedLaunch {
    int id;
    hipEvent_t a, b;
};

struct TimingCtx {  // per issuing thread
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    const std::atomic<uint32_t> *mask = nullptr;  // the handle's selection of kernels to time (bit = KernelId)
    bool open = false;                            // the launch in flight got a start event
};

struct HostScratch {  // per pool thread
    Delaunay dl;
    std::vector<int32_t> xy;
    std::vector<int32_t> sup;
    std::vector<int16_t> lattice;
    FilterScratch filter;  // latency mode: the lattice filters shared with the polling pool threads
};

// words of a pair's part of the device blob when the device lays it out itself (resident chunks): support points, two triangle lists, two vertex orders
static int blob_pair_words(const Dims &d) { return 3 * d.max_pts + 2 * 3 * d.max_tri + 2 * (d.max_pts + 1) + 64; }

// support points per pair fetched with the bulk D2H (more are fetched on demand): a sixth of the lattice, at least 4096
static int fsup_copy_pts(const Dims &d) { return std::min(d.max_pts, std::max(4096, d.Wc * d.Hc / 6)); }

struct Job {
    const uint8_t *left = nullptr, *right = nullptr;
    int batch = 0, stride = 0;
    float *d1 = nullptr, *d2 = nullptr;
    uint8_t *dmap = nullptr;  // host-memory jobs only: the caller wants the driver's 8-bit disparity image instead of the float maps
    int32_t *status = nullptr;
    int nchunks = 0;
    int issued2 = 0;  // chunks whose second GPU phase has been enqueued (guarded by sv_handle::mu)
    // host-memory jobs (sv_submit_batch_host): left/right/d1/d2 are HOST pointers; images go through the slot's device
    // staging buffers, maps come back from them.  Page-locked caller memory is the DMA source / target itself, pageable
    // memory goes through the slot's page-locked staging buffers.
    bool host = false, pin_in = false, pin_out = false;
    // single pairs in latency mode (run_inline) with page-locked memory: no copies at all - k_sobel reads the gray rows and the last
    // kernels write the maps over PCIe themselves (device-side addresses of the caller's buffers; nullptr: staged copies)
    const uint8_t *zc_left = nullptr, *zc_right = nullptr;
    float *zc_d1 = nullptr, *zc_d2 = nullptr;
    int drained = 0;  // chunks whose maps have reached the caller (guarded by sv_handle::mu)
    std::vector<hipEvent_t> ev_done;  // device-memory jobs: one event per chunk, recorded behind its second phase (sv_wait_batches waits on these, not on the streams)
};

enum SlotState { SLOT_FREE = 0, SLOT_BUSY = 1, SLOT_DRAINING = 2 };  // DRAINING: phase 2 issued, ev_free recorded

struct Slot {
    int id = 0;
    SlotDev dev{};
    int16_t *h_dcan = nullptr;  // pinned [cap][Hc*Wc]
    int32_t *h_fsup = nullptr;  // pinned [cap][fsup_copy_pts][3]: head of the GPU-filtered support lists
    int32_t *h_fnsup = nullptr; // pinned [cap]
    int32_t *h_blob = nullptr;  // pinned
    int16_t *h_dcan_dev = nullptr;  // the two page-locked buffers as the device addresses them (latency mode: copies by kernel)
    int32_t *h_blob_dev = nullptr;
    size_t blob_words = 0;
    hipEvent_t ev_p1 = nullptr, ev_free = nullptr, ev_sup = nullptr;
    // host-memory jobs: device staging (images in, maps out), page-locked staging for pageable callers, copy events
    uint8_t *d_in = nullptr;   // [2][cap][H][W]   gray rows, packed (left block, right block)
    float *d_out = nullptr;    // [2][cap][Nm]     final maps (left block, right block)
    uint8_t *h_in = nullptr;   // page-locked, same layout as d_in
    float *h_out = nullptr;    // page-locked, same layout as d_out
    uint8_t *d_out8 = nullptr, *h_out8 = nullptr;  // [cap][Nm] 8-bit disparity images of a chunk (device staging / page-locked mirror): dmap jobs
    hipEvent_t ev_in = nullptr, ev_lr = nullptr, ev_p2 = nullptr, ev_out = nullptr;
    // inputs of phase 1 for the chunk in flight (device pointers: the job's own, or the staging buffers)
    const uint8_t *in_left = nullptr, *in_right = nullptr;
    size_t in_pair = 0;
    int in_stride = 0;
    int gpu_pct = 0;            // share of this chunk's pairs whose triangulations go to the GPU kernel (set by the dispatcher)
    bool resident = false;      // this chunk's support lists stay on the device: preparation + triangulation by k_delaunay_resident in phase 1 (set by the issuer)
    bool grid_issued = false;   // latency mode: the candidate grid of this chunk was launched while the host still triangulated
    sv_handle *grid_h = nullptr;  // ... by early_grid_launch, possibly on a pool thread: its arguments
    size_t grid_off = 0;
    int grid_ns = 0;
    bool delivered_ok = false;  // host-memory jobs: the download succeeded (drainer -> deliverer)
    int up_ticket = -1, down_ticket = -1;  // DMA lanes: the chunk's image upload / map download in flight (DmaLanes tickets)
    bool out_enqueued = false;  // host-memory jobs: phase 2 was enqueued (ev_lr / ev_p2 are pending), the maps can be downloaded
    int state = SLOT_FREE;
    // chunk in flight
    Job *job = nullptr;
    int i0 = 0, n = 0;
    std::atomic<size_t> blob_off{0};
    std::atomic<int> pending{0};
    // latency mode (run_inline): the calling thread drives the chunk itself and spins on this flag instead of using the queues
    bool inline_mode = false;
    std::atomic<int> inline_done{0};
};

struct Task {
    Slot *s;
    int pair;  // index inside the chunk
    int side;  // -1: filter stage (then triangulates side 0 and queues side 1); 1: triangulation of the right image
    void (*fn)(void *) = nullptr;  // latency mode: a piece of a triangulation handed over by Delaunay::Spawn (then s == nullptr)
    void *arg = nullptr;
};


}  // namespace

struct sv_handle {
    sv_params p;
    sv_config cfg;
    KParams kp;      // image-space kernels (descriptor ... dense matching)
    KParams kp_map;  // map-space kernels (L/R check ... median): W/H/N = map size
    int nproc = 1;  // maps per pair that get post-processed
    int chunk = 1;
    bool block_sync = false;  // host waits on events sleep (throughput mode) instead of spinning (latency mode)
    int ns_margin_pct = 12;   // head room of the resident launch's LDS request over the largest set seen (SV_DG_MARGIN; 90: -7 % pairs/s, 25 -> 5: +0.7 %;
                              // a set beyond the request is handed to the pool, and the request grows at once)
    bool poll_sync = false;   // ... by asking the event and sleeping 40 us in between (wait_event)
    bool gpu_delaunay = false;  // divide-and-conquer phase of the triangulations on the GPU (delaunay_gpu.hip); the host only orders the vertices
    int gpu_delaunay_pct = 0;   // ... for this share of the pairs (100 in the GPU mode; a part in the host mode relieves the pool)
    int dg_sub_max = 0;           // vertex sets up to this size are triangulated whole in LDS, larger ones as subtrees of at most this size
    int dg_limit = 0;             // largest vertex set the GPU kernels take (LDS kernel: 4 000; with the slots' global-memory scratch: more)
    bool gpu_share_auto = false;  // host mode: the dispatcher moves that share up while the pool falls behind the GPU, down while it idles
    int auto_pct = 0, auto_acc = 0;  // (dispatcher thread only)
    bool share_sliced = false;    // the GPU kernel's share as a slice of every chunk instead of whole chunks
    int latency_split = 0;        // latency mode: depth of the triangulations' top levels shared with pool threads (0: none)
    int dbg_ccl_cap = 0, dbg_rt_cap = -1;  // sv_debug_set: overrides of the speckle stage's run-table size / the raster tile lists' size
    bool force_staging = false;            // sv_debug_set "host_force_staging"
    bool dbg_dma_fail = false;             // sv_debug_set "dma_selftest_fail"
    uint32_t dma_engines_override = 0;     // SV_DMA_ENGINES (experiments): engine of the upload lane | download lane << 8 | second download lane << 16, each as log2 + 1
    bool pool_sleep = false;               // sv_debug_set "pool_sleep"
    // latency mode, frames at a regular pace (a camera): the polling pool threads sleep between frames and are back shortly before the next one is due
    int64_t lat_last_start_ns = 0, lat_period_ns = 0;   // calling thread only
    std::atomic<int64_t> lat_next_expected_ns{0}, lat_period_ns_pub{0};
    int64_t lat_wake_lead_ns = 0;                        // 0: period / 16 within [0.3, 2] ms (SV_LAT_WAKE_LEAD_US)
    bool lat_filter_alone = false;         // latency mode: the lattice filters on the calling thread alone (sv_debug_set "lat_filter_alone")
    bool lat_runtime_copies = false;       // latency mode: the lattice / blob copies through hipMemcpyAsync as in the streamed path (sv_debug_set "lat_runtime_copies")
    int lat_pin = 0;                       // latency mode: keep the polling pool threads on the calling thread's L3 (sv_debug_set "latency_pin", SV_LATENCY_PIN)
    int lat_pin_l3 = -2;                   // the L3 domain they are pinned to right now (-2: never pinned)
    std::atomic<bool> lat_near{false};     // ... and they could be pinned there
    bool lat_auto = false;                 // sv_config.latency_split was 0: share the host stage only while lat_near
    cpu_set_t pool_cpus;                   // where the pool threads run otherwise (the GPU's NUMA node or the process's mask)
    // latency mode: a team call of the calling thread (team_run)
    std::atomic<int> team_open{0}, team_active{0}, team_next{0}, team_done{0};  // team_open: 0 or the number of the open call
    void (*team_fn)(void *, int) = nullptr;
    void *team_arg = nullptr;
    int team_parts = 0;
    int team_calls = 0;
    std::atomic<int> pollers{0};           // latency mode: pool threads polling the queue length right now
    bool resident_ok = false;     // the GPU's share of the chunks is built without the support lists ever leaving the device (k_delaunay_resident)
    std::atomic<int> shared_pct{0};   // host mode with a balanced share: the dispatcher's current share, read by the issuer (who decides per chunk)
    int issue_acc = 0;                // (issuer thread only) accumulator that turns the share into whole chunks
    std::atomic<int> ns_bound{0};     // vertices the resident kernel's LDS request is sized for: follows the support counts the chunks really have
    int resident_lds_max = 0;         // largest vertex set whose preparation + triangulation fit 160 KB of LDS at this image size (0: none - wide images)
    std::atomic<int64_t> gpu_tri_fallbacks{0};  // vertex sets of flagged pairs that the host triangulated after all (too large, or degenerate)
    std::atomic<int64_t> gpu_tri_pairs{0}, tri_pairs{0};  // pairs triangulated by the GPU kernel / all pairs, since creation
    bool node_bound = false;  // the handle's threads are bound to the CPUs of the GPU's NUMA node
    bool gpu_filter = false;  // lattice filters on the GPU (k_support_filter) instead of the host pool
    std::vector<Slot *> slots;
    int n_pf = 2;                                             // filter streams in use: 2; 4 where the triangulation chain of a 4K chunk (8 ms) follows the filter on them
    hipStream_t sP1 = nullptr, sPF[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // phase 1; lattice filter + its D2H (streams taken in turns:
                                                                // a 4K lattice keeps its one workgroup per pair busy for milliseconds)
    int pf_turn = 0;                                             // (issuer thread only)
    std::vector<hipStream_t> sP2;
    hipStream_t sIn = nullptr, sOut = nullptr, sOut2 = nullptr;  // host-memory jobs: image uploads / map downloads (two streams: two DMA engines), overlapping the kernels
    bool host_dev_ready = false, host_pin_in_ready = false, host_pin_out_ready = false;  // lazily allocated staging (guarded by host_mu)
    bool host_dev8_ready = false, host_pin_out8_ready = false;
    std::mutex host_mu;
    DmaLanes *dma = nullptr;   // engine-addressed SDMA copies of the host-memory path (dma_lanes.h); nullptr: hipMemcpyAsync on sIn / sOut / sOut2
    int host_copies_mode = 0;  // 0 not decided yet, 1 hipMemcpyAsync, 2 DMA lanes (SV_Q_HOST_COPIES)
    // control threads + queues
    std::thread t_issue, t_dispatch, t_finish, t_drain, t_deliver;
    std::mutex mu;  // guards: job queue + counters, quit, slot states, q1, q2, error
    std::condition_variable cv;
    std::deque<Job *> jobs;       // submitted, not yet picked up by the issuer
    std::vector<Job *> live;      // submitted, not yet waited for
    int jobs_submitted = 0, jobs_finished = 0;  // finished = all chunks' phase 2 enqueued
    int ring_pos = 0;             // next slot of the ring (continues across batches)
    bool quit = false;
    std::deque<Slot *> q1;  // phase 1 issued, waiting for the dispatcher
    std::deque<Slot *> q2;  // host stage complete, waiting for phase 2
    std::deque<Slot *> q3;  // host-memory jobs: phase 2 enqueued, waiting for the drainer (downloads the maps)
    std::deque<Slot *> q4;  // host-memory jobs: maps downloaded, waiting for the deliverer (pageable callers' copy, slot release)
    std::string error;
    std::atomic<bool> failed{false};
    std::vector<hipEvent_t> ev_pool;  // completion events of finished batches, reused (guarded by mu)
    // host pool
    std::vector<std::thread> pool;
    std::vector<HostScratch *> scratch;
    HostScratch *inline_scratch = nullptr;  // host-stage scratch of the calling thread (latency mode)
    bool inline_ok = true;                  // SV_NO_INLINE=1 (tests) sends single pairs through the queued pipeline as well
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<Task> queue;
    std::atomic<int> queue_len{0};  // == queue.size(); lets latency-mode pool threads poll for work without the lock
    bool pool_quit = false;
    // timing
    bool timing = false;
    std::atomic<uint32_t> timing_mask{0xFFFFFFFFu};  // kernels whose launches get events (sv_kernel_timing_select)
    TimingCtx tc_issue, tc_finish;
    std::mutex tmu;
    double k_ms[K_COUNT] = {0};
    int64_t k_calls[K_COUNT] = {0};
    std::atomic<int64_t> host_filter_ns{0}, host_delaunay_ns{0}, host_tasks{0};
    // debug
    std::map<std::string, std::vector<uint8_t>> dbg;
    unsigned long long *d_counters = nullptr;  // work counters of the matching kernels (sv_debug_counters)
    uint8_t *dbg_desc = nullptr;               // keep_debug: descriptor images [cap][2][N][16] for the stage snapshot
    bool lat_trace = false;                    // SV_LAT_TRACE=1: wall-clock split of the latency path, printed by sv_destroy
    double issue_ns[6] = {0, 0, 0, 0, 0, 0};   // lat_trace: issuer wall time in sobel+support launches / filter launches / triangulation launch / copies / event records / waiting for a slot
    long issue_chunks = 0;
    double lat_ns[8] = {0};
    double lat_sub_ns[3] = {0};  // lat_trace, inside "filter + left triangulation": lattice filters, hand-over (grid upload + launch, right-side task), left triangulation
    long lat_calls = 0;
    double drain_ns[3] = {0};  // drainer, per chunk: waiting for phase 2, downloading, handing pageable maps over
    long drain_chunks = 0;
    // SV_CHUNK_TRACE=<file>: wall-clock time of every pipeline stage of every chunk (tools/chunk_trace.py reads it), written by sv_destroy
    struct TraceRec {
        int slot, stage;
        int64_t ns;
    };
    bool chunk_trace_on = false;
    std::string chunk_trace_path;
    std::mutex trace_mu;
    std::vector<TraceRec> chunk_trace;
    std::chrono::steady_clock::time_point t_created = std::chrono::steady_clock::now();
};

namespace {

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            char buf_[512];                                                                                   \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            throw std::runtime_error(buf_);                                                                   \
        }                                                                                                     \
    } while (0)

enum TraceStage { TR_SLOT = 0, TR_UP_QUEUED, TR_UP_DONE, TR_P1_QUEUED, TR_P1_DONE, TR_HOST_DONE, TR_P2_QUEUED, TR_P2_DONE, TR_DOWN_QUEUED, TR_DOWN_DONE, TR_FREE, TR_COUNT };
const char *const kTraceStageNames[TR_COUNT] = {"slot", "upload_queued", "upload_done", "phase1_queued", "phase1_done", "host_done", "phase2_queued", "phase2_done", "download_queued",
                                                "download_done", "free"};

inline void trace_mark(sv_handle *h, const Slot *s, int stage) {
    if (!h->chunk_trace_on) return;
    const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - h->t_created).count();
    std::lock_guard<std::mutex> lk(h->trace_mu);
    h->chunk_trace.push_back({s->id, stage, ns});
}

// The only place this file reads the environment.
bool env_int(const char *name, int *out) {
    const char *e = getenv(name);
    if (!e || !*e) return false;
    if (out) *out = !strcmp(e, "block") ? 1 : (!strcmp(e, "spin") ? 2 : (!strcmp(e, "poll") ? 3 : atoi(e)));
    return true;
}

// Environment variables override the policy fields of sv_config (include/stereo_vision_hip.h names the variable beside each field):
// a deployment can be steered without a rebuild, and everything that steers a handle is visible in ONE struct afterwards.
void apply_env_overrides(sv_config &c) {
    int v;
    if (env_int("SV_GPU_FILTER", &v) && v) c.gpu_lattice_filter = 1;
    if (env_int("SV_HOST_FILTER", &v) && v) c.gpu_lattice_filter = 2;
    if (env_int("SV_GPU_DELAUNAY", &v)) {  // 1: all on the GPU; 0: never by the handle's own choice (balanced even with few host threads)
        if (v) c.gpu_triangulation = 1;
        else if (c.gpu_triangulation == 0 || c.gpu_triangulation == 1) c.gpu_triangulation = 4;
    }
    if (env_int("SV_GPU_DELAUNAY_PCT", &v)) c.gpu_triangulation = 3, c.gpu_triangulation_pct = std::max(0, std::min(100, v));
    if (env_int("SV_GPU_DELAUNAY_AUTO", &v) && !v && (c.gpu_triangulation == 0 || c.gpu_triangulation == 4)) c.gpu_triangulation = 2;
    if (env_int("SV_GPU_DELAUNAY_SLICED", &v)) c.share_sliced = 1;
    if (env_int("SV_GPU_DELAUNAY_MAX", &v)) c.dg_max_points = std::max(16, v);
    if (env_int("SV_DG_SUBMAX", &v)) c.dg_sub_max = std::max(6, v);
    if (env_int("SV_RESIDENT", &v)) c.resident = v ? 0 : 2;
    if (env_int("SV_NO_AFFINITY", &v)) c.affinity = 2;
    if (env_int("SV_NO_INLINE", &v)) c.inline_latency_path = 2;
    if (env_int("SV_LATENCY_SPLIT", &v)) c.latency_split = v >= 0 && v <= 3 ? v : 0;
    if (env_int("SV_EVENT_SYNC", &v)) c.event_sync = v == 1 ? 1 : (v == 3 ? 3 : 2);
    if (env_int("SV_HOST_COPIES", &v)) c.host_copies = v == 1 ? 1 : (v == 2 ? 2 : 0);
}

int validate(const sv_params &p, const sv_config &c, std::string &err) {
    char b[256];
    if (p.disp_max < 10 || p.disp_max > 1023) {
        err = "disp_max must be in [10,1023]";
        return SV_ERR_ARG;
    }
    if (c.width < 32 || c.height < 32 || c.width > 8192 || c.height > 4096) {
        snprintf(b, sizeof(b), "unsupported image size %dx%d (32..8192 x 32..4096)", c.width, c.height);
        err = b;
        return SV_ERR_ARG;
    }
    if (p.candidate_stepsize < 1 || p.grid_size < 1 || p.incon_window_size < 0) {
        err = "bad lattice / grid parameters";
        return SV_ERR_ARG;
    }
    const float pr = fmaxf((float)ceil(p.sigma * p.sradius), 2.0f);
    if (!(pr <= 15.0f)) {
        err = "plane radius ceil(sigma*sradius) must be <= 15";
        return SV_ERR_ARG;
    }
    if ((size_t)c.width * c.height >= (1u << 30)) {
        err = "image too large";
        return SV_ERR_ARG;
    }
    // policy words: a caller built against an older, shorter sv_config hands over garbage here - refuse it instead of acting on it
    const struct {
        const char *name;
        int32_t v, lo, hi;
    } fields[] = {{"device", c.device, 0, 1023}, {"n_workers", c.n_workers, 0, 1024}, {"chunk", c.chunk, 0, 4096}, {"n_streams", c.n_streams, 0, 64}, {"n_slots", c.n_slots, 0, 64},
                  {"gpu_lattice_filter", c.gpu_lattice_filter, 0, 2}, {"gpu_triangulation", c.gpu_triangulation, 0, 4}, {"gpu_triangulation_pct", c.gpu_triangulation_pct, 0, 100},
                  {"resident", c.resident, 0, 2}, {"dg_sub_max", c.dg_sub_max, 0, 1 << 20}, {"dg_max_points", c.dg_max_points, 0, 1 << 24}, {"affinity", c.affinity, 0, 2},
                  {"inline_latency_path", c.inline_latency_path, 0, 2}, {"event_sync", c.event_sync, 0, 3}, {"share_sliced", c.share_sliced, 0, 1}, {"latency_split", c.latency_split, 0, 3},
                  {"host_copies", c.host_copies, 0, 2}};
    for (const auto &f : fields)
        if (f.v < f.lo || f.v > f.hi) {
            snprintf(b, sizeof(b), "sv_config.%s = %d is outside [%d, %d] (is the caller built against this header's sv_config: %d int32 words?)", f.name, (int)f.v, (int)f.lo, (int)f.hi,
                     (int)(sizeof(sv_config) / sizeof(int32_t)));
            err = b;
            return SV_ERR_ARG;
        }
    for (int32_t r : c.reserved)
        if (r != 0) {
            err = "sv_config.reserved must be 0";
            return SV_ERR_ARG;
        }
    return SV_OK;
}

void fill_kparams(sv_handle *h) {
    const sv_params &p = h->p;
    KParams &k = h->kp;
    Dims &d = k.d;
    d.W = h->cfg.width;
    d.H = h->cfg.height;
    d.N = d.W * d.H;
    d.step = p.candidate_stepsize;
    d.sub = p.subsampling ? 1 : 0;
    if (d.sub) d.step += d.step % 2;  // elas.cpp:376-378: an even lattice step at half resolution
    d.Wm = d.sub ? d.W / 2 : d.W;     // elas.h:160-161: half-size maps
    d.Hm = d.sub ? d.H / 2 : d.H;
    d.Nm = d.Wm * d.Hm;
    d.Wc = (d.W + d.step - 1) / d.step;
    d.Hc = (d.H + d.step - 1) / d.step;
    d.grid_size = p.grid_size;
    d.gw = (int)ceil((float)d.W / (float)p.grid_size);  // elas.cpp:88-89
    d.gh = (int)ceil((float)d.H / (float)p.grid_size);
    d.ncell = d.gw * d.gh;
    d.disp_max = p.disp_max;
    d.disp_min = std::max(p.disp_min, 0);  // elas.cpp:318: only the support matching's range starts there
    d.D = p.disp_max + 1;
    d.MW = (d.D + 31) / 32;
    d.max_pts = (d.Wc - 1) * (d.Hc - 1) + 6;
    d.max_tri = 2 * d.max_pts;
    k.support_threshold = p.support_threshold;
    k.support_texture = p.support_texture;
    k.lr_threshold = p.lr_threshold;
    k.match_texture = p.match_texture;
    // elas.cpp:828-832.  expf/logf on float arguments, as the reference's exp()/log() calls resolve.
    const float two_sigma_squared = 2 * p.sigma * p.sigma;
    k.plane_radius = (int32_t)fmaxf((float)ceil(p.sigma * p.sradius), (float)2.0);
    for (int delta_d = 0; delta_d < 16; delta_d++)
        k.prior[delta_d] = (int32_t)((-logf(p.gamma + expf(-delta_d * delta_d / two_sigma_squared)) + logf(p.gamma)) / p.beta);
    k.speckle_sim = p.speckle_sim_threshold;
    k.speckle_size = p.speckle_size;
    k.gap_width = p.ipol_gap_width;
    k.add_corners = p.add_corners;
    // run tables of the speckle stage (LDS of k_ccl_band, 12 B per run next to the band's bit masks).  Real disparity maps are
    // fragmented: kitti_mini pair 0 has up to 170 runs per row, i.e. 1 400 per 8-row band.  4096 runs per band (512 per row) for
    // images up to 2048 columns, more for wider ones, within 144 KB of the CU's 160 KB; beyond that a map takes the slow path.
    k.ccl_cap = std::max(4096, std::min(8192, ((d.W * 2 + 1023) / 1024) * 1024));
    if (h->dbg_ccl_cap > 0) k.ccl_cap = h->dbg_ccl_cap;  // tests (sv_debug_set "ccl_cap"): force the per-pixel slow path
    {
        const long room = (144L * 1024 - 256 - 8L * ((d.W + 63) / 64) * 28) / 8;  // (two words of LDS per run: kernels.hip ccl_lds_bytes)
        k.ccl_cap = (int)std::max(1L, std::min((long)k.ccl_cap, room));
    }
    {  // multiply-shift for the grid column and row of a pixel (k_dense): valid only if it reproduces the division for every column and row
        const uint32_t gs = (uint32_t)p.grid_size, m = (65536u + gs - 1u) / gs;
        bool ok = m < (1u << 24) && d.W <= (1 << 16) && d.H <= (1 << 16);
        for (uint32_t u = 0; ok && u < (uint32_t)std::max(d.W, d.H); u++) ok = ((u * m) >> 16) == u / gs && (int)floorf((float)u / (float)gs) == (int)(u / gs);
        k.cell_mul = ok ? m : 0u;
    }
    k.rt_cap = 512;
    if (h->dbg_rt_cap >= 0) k.rt_cap = std::min(512, h->dbg_rt_cap);  // tests (sv_debug_set "rt_cap"): force the raster fallback
    // the post-matching stages see the map as their image: W/H/N are the map size, and at half resolution the speckle and
    // gap limits shrink (elas.cpp:1017-1022, 1130-1135)
    KParams &km = h->kp_map;
    km = k;
    km.d.W = d.Wm;
    km.d.H = d.Hm;
    km.d.N = d.Nm;
    if (d.sub) {
        km.speckle_size = (int32_t)(sqrt((float)p.speckle_size) * 2);
        km.gap_width = p.ipol_gap_width / 2 + 1;
    }
    h->nproc = p.postprocess_only_left ? 1 : 2;
    // the on-GPU lattice filter unrolls the 11 x 11 window and resolves a point's earlier neighbours with one 64-lane ballot:
    // needs incon_window_size <= 5 (any lattice size: its state lives in global memory)
    h->gpu_filter = h->cfg.gpu_lattice_filter != 2 && p.incon_window_size >= 0 && p.incon_window_size <= 5 &&
                    p.incon_min_support >= 0 && p.incon_min_support <= 60 && p.incon_threshold >= 0 && p.incon_threshold < 4096;  // c_late lives in 6 bits; the classify kernel's sentinel is 16384
    // (sv_create additionally keeps the filters on the host for chunk < 4: the GPU version is a ~0.4 ms latency chain,
    //  worth it only when many pairs share it)
}

void timing_hook(void *ctx, int id, bool before, hipStream_t st) {
    TimingCtx *t = (TimingCtx *)ctx;
    if (before) {
        t->open = false;
        if (t->mask && !((t->mask->load(std::memory_order_relaxed) >> id) & 1u)) return;
        if (t->used + 2 > t->pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            t->pool.push_back(a);
            t->pool.push_back(b);
        }
        TimedLaunch tl{id, t->pool[t->used], t->pool[t->used + 1]};
        t->used += 2;
        t->timed.push_back(tl);
        t->open = true;
        (void)hipEventRecord(tl.a, st);
    } else if (t->open) {
        t->open = false;
        (void)hipEventRecord(t->timed.back().b, st);
    }
}

void collect_timing(sv_handle *h, TimingCtx *t) {
    if (t->timed.empty()) return;
    std::lock_guard<std::mutex> g(h->tmu);
    for (const TimedLaunch &tl : t->timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, tl.a, tl.b) == hipSuccess) {
            h->k_ms[tl.id] += ms;
            h->k_calls[tl.id] += 1;
        }
    }
    t->timed.clear();
    t->used = 0;
}

void note_error(sv_handle *h, const char *what) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->failed) {
        h->failed = true;
        h->error = what;
    }
}

// ---- debug snapshots (keep_debug: one slot, one pair per chunk) -----------------------------------------------------
template <class T>
void dbg_put(sv_handle *h, const char *name, const T *data, size_t count) {
    std::vector<uint8_t> &v = h->dbg[name];
    v.resize(count * sizeof(T));
    if (count) memcpy(v.data(), data, count * sizeof(T));
}

void dbg_from_device(sv_handle *h, hipStream_t st, const char *name, const void *dptr, size_t bytes) {
    std::vector<uint8_t> &v = h->dbg[name];
    v.resize(bytes);
    HIP_TRY(hipStreamSynchronize(st));
    if (bytes) HIP_TRY(hipMemcpy(v.data(), dptr, bytes, hipMemcpyDeviceToHost));
}

void dbg_maps(sv_handle *h, hipStream_t st, const char *stage, const float *base, int j) {
    const size_t N = h->kp.d.Nm;
    char name[64];
    for (int side = 0; side < 2; side++) {
        snprintf(name, sizeof(name), "%s%d", stage, side + 1);
        dbg_from_device(h, st, name, base + ((size_t)j * 2 + side) * N, N * sizeof(float));
    }
}

// 16-bit device maps, reported as f32 like every other stage
void dbg_maps_i16(sv_handle *h, hipStream_t st, const char *stage, const int16_t *base, int j) {
    const size_t N = h->kp.d.Nm;
    char name[64];
    std::vector<int16_t> raw(N);
    HIP_TRY(hipStreamSynchronize(st));
    for (int side = 0; side < 2; side++) {
        snprintf(name, sizeof(name), "%s%d", stage, side + 1);
        HIP_TRY(hipMemcpy(raw.data(), base + ((size_t)j * 2 + side) * N, N * sizeof(int16_t), hipMemcpyDeviceToHost));
        std::vector<uint8_t> &v = h->dbg[name];
        v.resize(N * sizeof(float));
        float *f = reinterpret_cast<float *>(v.data());
        for (size_t i = 0; i < N; i++) f[i] = (float)raw[i];
    }
}

// maps after an out-of-place stage: processed sides live in `cur`; with postprocess_only_left the right map stays in `disp`
void dbg_maps_nproc(sv_handle *h, hipStream_t st, const char *stage, const float *cur, const float *disp, int j) {
    const size_t N = h->kp.d.Nm;
    char name[64];
    for (int side = 0; side < 2; side++) {
        snprintf(name, sizeof(name), "%s%d", stage, side + 1);
        const float *base = (side < h->nproc) ? cur : disp;
        dbg_from_device(h, st, name, base + ((size_t)j * 2 + side) * N, N * sizeof(float));
    }
}

// compacted candidate lists in the reference's layout (elas.cpp:631-648) from the device bit masks
void dbg_grid(sv_handle *h, hipStream_t st, Slot *s, int j) {
    const Dims &d = h->kp.d;
    std::vector<uint32_t> m((size_t)d.ncell * d.MW);
    for (int side = 0; side < 2; side++) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(m.data(), s->dev.gmaskB + ((size_t)j * 2 + side) * d.ncell * d.MW, m.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> g((size_t)d.ncell * (d.disp_max + 2), 0);
        for (int c = 0; c < d.ncell; c++) {
            int cnt = 0;
            for (int dd = 0; dd <= d.disp_max; dd++)
                if ((m[(size_t)c * d.MW + (dd >> 5)] >> (dd & 31)) & 1u) g[(size_t)c * (d.disp_max + 2) + (++cnt)] = dd;
            g[(size_t)c * (d.disp_max + 2)] = cnt;
        }
        dbg_put(h, side ? "grid2" : "grid1", g.data(), g.size());
    }
    int32_t gd[3] = {d.disp_max + 2, d.gw, d.gh};
    dbg_put(h, "grid_dims", gd, 3);
}

// A host thread waits for an event of the pipeline.
hipError_t wait_event(const sv_handle *h, hipEvent_t ev) {
    if (!h->poll_sync) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t r = hipEventQuery(ev);
        if (r != hipErrorNotReady) return r;
        const struct timespec ts = {0, 40000};
        nanosleep(&ts, nullptr);
    }
}

// ---- stage 1: issuer ---------------------------------------------------------------------------------------------------
void issue_phase1(sv_handle *h, Slot *s) {
    const KParams &k = h->kp;
    const Dims &d = k.d;
    const int lat = d.Wc * d.Hc;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int what) {  // lat_trace: where the issuing thread's time goes
        if (!h->lat_trace) return;
        const auto now = std::chrono::steady_clock::now();
        h->issue_ns[what] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t_last).count();
        t_last = now;
    };
    launch_sobel(k, s->in_left, s->in_right, s->in_pair, s->in_stride, s->dev, s->n, h->sP1);
    launch_support(k, s->dev, s->n, h->sP1);
    lap(0);
    hipStream_t tail = h->sP1;
    if (h->gpu_filter) {
        // the filter is one long-running workgroup per pair: on its own stream it does not hold up the next chunk's phase 1
        HIP_TRY(hipEventRecord(s->ev_sup, h->sP1));
        tail = h->sPF[h->pf_turn];
        h->pf_turn = (h->pf_turn + 1) % h->n_pf;
        HIP_TRY(hipStreamWaitEvent(tail, s->ev_sup, 0));
        launch_support_filter(k, h->p.incon_window_size, h->p.incon_threshold, h->p.incon_min_support, s->dev, s->n, tail);
        lap(1);
        if (s->resident) {
            // the support lists stay where they are: preparation and triangulation of both sides of every pair in one launch, straight
            // from the filter's buffers into the blob (laid out by the kernel); only the meta words come back - counts for the launch
            // sizes of phase 2 and the callers' status, and a triangle count of -1 for a side the host has to build (coincident points)
            // (the launch requests LDS for sets of up to `bound` vertices and hands larger ones back: the bound follows the chunks' real counts)
            const int bound = std::min({h->ns_bound.load(std::memory_order_relaxed), h->dg_sub_max, h->resident_lds_max});
            const bool large = s->dev.dg.prep != nullptr;  // sets beyond LDS (4K lattices): prepared in global memory, triangulated by the cut path
            launch_delaunay_resident(s->dev.fsup, s->dev.fnsup, s->dev.blob, s->dev.cap, d.max_pts, blob_pair_words(d), s->n, bound, bound, d.W, d.H, d.step, d.disp_max, tail,
                                     large ? h->dg_sub_max : 0x7FFFFFFF, large ? h->dg_limit : 0);
            if (large) launch_delaunay_resident_large(s->dev.blob, s->n, h->dg_sub_max, h->dg_limit, d.W, d.H, d.step, d.disp_max, s->dev.dg.prep, s->dev.dg, tail);
            lap(2);
            HIP_TRY(hipMemcpyAsync(s->h_blob, s->dev.blob, sizeof(int32_t) * META_WORDS * (size_t)s->n, hipMemcpyDeviceToHost, tail));
        } else {
            HIP_TRY(hipMemcpyAsync(s->h_fnsup, s->dev.fnsup, sizeof(int32_t) * (size_t)s->n, hipMemcpyDeviceToHost, tail));
            const size_t w = sizeof(int32_t) * 3 * (size_t)fsup_copy_pts(d);
            HIP_TRY(hipMemcpy2DAsync(s->h_fsup, w, s->dev.fsup, sizeof(int32_t) * 3 * (size_t)d.max_pts, w, (size_t)s->n,
                                     hipMemcpyDeviceToHost, tail));
        }
    }
    if (!h->gpu_filter || h->cfg.keep_debug) {
        if (s->inline_mode && s->h_dcan_dev && !h->lat_runtime_copies)
            launch_copy_block(s->h_dcan_dev, s->dev.dcan, sizeof(int16_t) * (size_t)s->n * lat, tail);  // single pair: in stream order, by a kernel
        else
            HIP_TRY(hipMemcpyAsync(s->h_dcan, s->dev.dcan, sizeof(int16_t) * (size_t)s->n * lat, hipMemcpyDeviceToHost, tail));
    }
    lap(3);
    HIP_TRY(hipGetLastError());  // a rejected phase-1 launch must not leave the previous chunk's lattices to the host stage
    HIP_TRY(hipEventRecord(s->ev_p1, tail));
    lap(4);
    if (h->lat_trace) h->issue_chunks++;
    if (h->cfg.keep_debug) {
        const int j = s->n - 1;
        HIP_TRY(hipStreamSynchronize(tail));
        // the descriptor images as the reference would hold them: assembled from the gradient planes by the same device
        // function the matching kernels use (they are never stored otherwise)
        launch_expand_debug(k, s->dev, s->n, h->dbg_desc, h->sP1);
        dbg_from_device(h, h->sP1, "desc1", h->dbg_desc + ((size_t)j * 2) * d.N * 16, (size_t)d.N * 16);
        dbg_from_device(h, h->sP1, "desc2", h->dbg_desc + ((size_t)j * 2 + 1) * d.N * 16, (size_t)d.N * 16);
        {
            std::vector<int16_t> rowmajor((size_t)lat);  // the oracle's layout is [Hc][Wc]; the device writes [Wc][Hc]
            const int16_t *T = s->h_dcan + (size_t)j * lat;
            for (int vc = 0; vc < d.Hc; vc++)
                for (int uc = 0; uc < d.Wc; uc++) rowmajor[(size_t)vc * d.Wc + uc] = T[(size_t)uc * d.Hc + vc];
            dbg_put(h, "dcan_raw", rowmajor.data(), lat);
        }
        int32_t dd[2] = {d.Wc, d.Hc};
        dbg_put(h, "dcan_dims", dd, 2);
    }
}

// ---- host-memory jobs: staging and copies ------------------------------------------------------------------------------
void spawn_to_pool(void *ctx, void (*fn)(void *), void *arg);
int latency_pollers(const sv_handle *h);

// A large host-side copy shared between the calling thread and idle pool threads: pieces are claimed from an atomic counter,
// so the caller never waits for a helper that is busy with a triangulation (it just ends up copying more itself).
struct CopyJob {
    struct Piece {
        void *dst;
        const void *src;
        size_t bytes;
    };
    std::vector<Piece> pieces;
    std::atomic<int> next{0}, done{0}, refs{1};
    static void work(CopyJob *c) {
        const int n = (int)c->pieces.size();
        for (;;) {
            const int i = c->next.fetch_add(1);
            if (i >= n) break;
            memcpy(c->pieces[i].dst, c->pieces[i].src, c->pieces[i].bytes);
            c->done.fetch_add(1, std::memory_order_release);
        }
    }
    static void helper(void *arg) {
        CopyJob *c = static_cast<CopyJob *>(arg);
        work(c);
        if (c->refs.fetch_sub(1) == 1) delete c;
    }
};

constexpr size_t COPY_PIECE = (size_t)512 << 10;

struct CopyList {  // collects (dst, src, bytes) ranges, cut into pieces of at most `piece` bytes
    CopyJob *job = new CopyJob();
    size_t piece = COPY_PIECE;
    void add(void *dst, const void *src, size_t bytes) {
        for (size_t o = 0; o < bytes; o += piece)
            job->pieces.push_back({static_cast<uint8_t *>(dst) + o, static_cast<const uint8_t *>(src) + o, std::min(piece, bytes - o)});
    }
    // runs the copies; returns when every byte has landed
    void run(sv_handle *h, int max_helpers) {
        CopyJob *c = job;
        job = nullptr;
        const int n = (int)c->pieces.size();
        const int helpers = std::max(0, std::min({max_helpers, n - 1, (int)h->pool.size()}));
        c->refs.store(1 + helpers);
        for (int i = 0; i < helpers; i++) spawn_to_pool(h, CopyJob::helper, c);
        CopyJob::work(c);
        while (c->done.load(std::memory_order_acquire) < n) __builtin_ia32_pause();
        if (c->refs.fetch_sub(1) == 1) delete c;
    }
    ~CopyList() { delete job; }
};

// true when [p, p + bytes) is page-locked host memory the device can DMA from / to (hipHostMalloc, hipHostRegister, sv_host_alloc)
bool is_pinned_host(const void *p, size_t bytes) {
    if (!p || bytes == 0) return false;
    for (const uint8_t *q : {static_cast<const uint8_t *>(p), static_cast<const uint8_t *>(p) + bytes - 1}) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof(a));
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();  // plain pageable memory: the query fails and leaves a sticky error behind
            return false;
        }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Staging of the host-memory path, allocated at the first host job (device-memory users never pay for it): per slot the
// packed gray images of a chunk and its two final maps on the device, and - only when a caller hands over pageable memory -
// page-locked mirrors of both.
void ensure_host_staging(sv_handle *h, bool need_pin_in, bool need_pin_out, bool dmap = false) {
    std::lock_guard<std::mutex> lk(h->host_mu);
    const Dims &d = h->kp.d;
    const size_t cap = (size_t)h->chunk;
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (!h->host_dev_ready) {
        HIP_TRY(hipStreamCreateWithFlags(&h->sIn, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->sOut, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&h->sOut2, hipStreamNonBlocking));
        const unsigned evf = hipEventDisableTiming | (h->block_sync ? hipEventBlockingSync : 0u);
        for (Slot *sl : h->slots) {
            HIP_TRY(hipMalloc((void **)&sl->d_in, 2 * cap * (size_t)d.N));
            HIP_TRY(hipMalloc((void **)&sl->d_out, 2 * cap * (size_t)d.Nm * sizeof(float)));
            HIP_TRY(hipEventCreateWithFlags(&sl->ev_in, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sl->ev_lr, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sl->ev_p2, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sl->ev_out, evf));
        }
        // Who moves the chunks over PCIe in the pipeline: SDMA engines of our own choice - one for the image uploads, another for the
        // map downloads - where the runtime offers engine-addressed copies, else hipMemcpyAsync on the three copy streams (which picks
        // "the first engine that is free" per copy: the directions end up on one engine every few chunks and serialise).  Single pairs
        // on a chunk-1 handle (run_inline) copy in stream order on the kernels' own streams either way.
        h->host_copies_mode = 1;
        if (h->cfg.host_copies != 1) {
            std::string why;
            h->dma = DmaLanes::create(h->slots[0]->d_in, h->slots[0]->h_blob, &why, h->dma_engines_override);
            if (h->dma) {
                // every lane moves a few bytes both ways before a batch depends on it (an engine id the runtime takes at creation but
                // refuses - or never completes - for this pair of agents would otherwise fail the first batch): the slot's page-locked
                // blob and its image staging are idle here
                Slot *s0 = h->slots[0];
                int32_t *hb = s0->h_blob;
                const int32_t keep[2] = {hb[0], hb[1]};
                hb[0] = 0x5356444D, hb[1] = 0x414C414E;  // "SVDM" "ALAN"
                bool ok = true;
                for (DmaLanes::Lane lane : {DmaLanes::DOWN, DmaLanes::DOWN2}) {
                    const int t = h->dma->begin(1);
                    ok = ok && t >= 0 && h->dma->add(t, DmaLanes::UP, s0->d_in, hb, 8, true) && h->dma->wait(t, false, 2000);
                    hb[0] = hb[1] = 0;
                    const int t2 = ok ? h->dma->begin(1) : -1;
                    ok = ok && t2 >= 0 && h->dma->add(t2, lane, hb, s0->d_in, 8, false) && h->dma->wait(t2, false, 2000) && hb[0] == 0x5356444D && hb[1] == 0x414C414E;
                }
                hb[0] = keep[0], hb[1] = keep[1];
                if (h->dbg_dma_fail) ok = false;
                if (!ok) {
                    why = "the SDMA lanes failed their self-test (" + h->dma->describe() + ")";
                    delete h->dma;
                    h->dma = nullptr;
                }
            }
            if (h->dma) h->host_copies_mode = 2;
            else if (h->cfg.host_copies == 2) throw std::runtime_error("sv_config.host_copies = 2, but " + why);
        }
        h->host_dev_ready = true;
    }
    if (need_pin_in && !h->host_pin_in_ready) {
        for (Slot *sl : h->slots) HIP_TRY(hipHostMalloc((void **)&sl->h_in, 2 * cap * (size_t)d.N, hipHostMallocDefault));
        h->host_pin_in_ready = true;
    }
    if (need_pin_out && !dmap && !h->host_pin_out_ready) {
        for (Slot *sl : h->slots) HIP_TRY(hipHostMalloc((void **)&sl->h_out, 2 * cap * (size_t)d.Nm * sizeof(float), hipHostMallocDefault));
        h->host_pin_out_ready = true;
    }
    if (dmap && !h->host_dev8_ready) {
        for (Slot *sl : h->slots) HIP_TRY(hipMalloc((void **)&sl->d_out8, cap * (size_t)d.Nm));
        h->host_dev8_ready = true;
    }
    if (dmap && need_pin_out && !h->host_pin_out8_ready) {
        for (Slot *sl : h->slots) HIP_TRY(hipHostMalloc((void **)&sl->h_out8, cap * (size_t)d.Nm, hipHostMallocDefault));
        h->host_pin_out8_ready = true;
    }
}

// Upload of a chunk's images on `st` (stream sIn in the pipeline, the phase-1 stream itself in latency mode).  Device layout:
// rows packed to W bytes, all left images of the chunk, then all right images.
// lanes (pipeline only): the copies go to the upload engine of h->dma instead of `st` and s->up_ticket names them - whoever launches
// phase 1 waits for the ticket on the host first (the issuer keeps the next chunks' uploads queued on the engine meanwhile).
// Pageable images of a chunk packed into the slot's page-locked mirror (both sides), by this thread and `copy_helpers` pool threads
void pack_images(sv_handle *h, Slot *s, int copy_helpers, size_t piece) {
    const Dims &d = h->kp.d;
    const Job &job = *s->job;
    const size_t cap = (size_t)s->dev.cap, img = (size_t)d.N, src_pair = (size_t)d.H * job.stride;
    CopyList cl;
    cl.piece = piece;
    for (int side = 0; side < 2; side++) {
        const uint8_t *src = (side ? job.right : job.left) + (size_t)s->i0 * src_pair;
        uint8_t *stage = s->h_in + side * cap * img;
        if (job.stride == d.W) {
            cl.add(stage, src, (size_t)s->n * img);
        } else {
            for (int j = 0; j < s->n; j++)
                for (int y = 0; y < d.H; y++) cl.job->pieces.push_back({stage + (size_t)j * img + (size_t)y * d.W, src + (size_t)j * src_pair + (size_t)y * job.stride, (size_t)d.W});
        }
    }
    cl.run(h, copy_helpers);
}

void upload_chunk(sv_handle *h, Slot *s, hipStream_t st, int copy_helpers, bool lanes = false) {
    const Dims &d = h->kp.d;
    const Job &job = *s->job;
    const size_t cap = (size_t)s->dev.cap, img = (size_t)d.N, src_pair = (size_t)d.H * job.stride;
    const uint8_t *src[2] = {job.left + (size_t)s->i0 * src_pair, job.right + (size_t)s->i0 * src_pair};
    s->up_ticket = -1;
    if (lanes && h->dma && !(job.pin_in && job.stride != d.W)) {  // (strided page-locked rows stay with the runtime's 2-D copy)
        s->up_ticket = h->dma->begin(2);
        if (s->up_ticket < 0) throw std::runtime_error("DMA lanes: no completion signal");
    }
    for (int side = 0; side < 2; side++) {
        uint8_t *dev = s->d_in + side * cap * img;
        if (s->up_ticket >= 0) {
            const uint8_t *from = src[side];
            if (!job.pin_in) {  // pageable: pack into the page-locked mirror first (the left images travel while the right ones are packed)
                uint8_t *stage = s->h_in + side * cap * img;
                CopyList cl;
                if (job.stride == d.W) {
                    cl.add(stage, src[side], (size_t)s->n * img);
                } else {
                    for (int j = 0; j < s->n; j++)
                        for (int y = 0; y < d.H; y++) cl.job->pieces.push_back({stage + (size_t)j * img + (size_t)y * d.W, src[side] + (size_t)j * src_pair + (size_t)y * job.stride, (size_t)d.W});
                }
                cl.run(h, copy_helpers);
                from = stage;
            }
            if (!h->dma->add(s->up_ticket, DmaLanes::UP, dev, from, (size_t)s->n * img, true)) note_error(h, "DMA lanes: an image upload was refused");
            continue;
        }
        if (job.pin_in) {  // DMA straight from the caller's page-locked memory
            if (job.stride == d.W)
                HIP_TRY(hipMemcpyAsync(dev, src[side], (size_t)s->n * img, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync(dev, (size_t)d.W, src[side], (size_t)job.stride, (size_t)d.W, (size_t)s->n * d.H, hipMemcpyHostToDevice, st));
            continue;
        }
        uint8_t *stage = s->h_in + side * cap * img;  // pageable: pack into the page-locked mirror first, one image side at a
        CopyList cl;                                  // time so that the DMA of the left images overlaps the packing of the right
        if (job.stride == d.W) {
            cl.add(stage, src[side], (size_t)s->n * img);
        } else {
            for (int j = 0; j < s->n; j++)
                for (int y = 0; y < d.H; y++) cl.job->pieces.push_back({stage + (size_t)j * img + (size_t)y * d.W, src[side] + (size_t)j * src_pair + (size_t)y * job.stride, (size_t)d.W});
        }
        cl.run(h, copy_helpers);
        HIP_TRY(hipMemcpyAsync(dev, stage, (size_t)s->n * img, hipMemcpyHostToDevice, st));
    }
    s->in_left = s->d_in;
    s->in_right = s->d_in + cap * img;
    s->in_pair = img;
    s->in_stride = d.W;
}

// Downloads of one finished map block (side 0: left maps, 1: right maps) of a chunk, as a list of copies.  Pairs with fewer than 3
// support points are skipped: the reference leaves the caller's maps untouched for them (elas.cpp:63-69).  `second`: the copy may go
// to the second download stream / engine (long runs go down as two halves).
struct DlCopy {
    void *dst;
    const void *src;
    size_t bytes;
    bool second;
};

void plan_downloads(sv_handle *h, Slot *s, int side, bool split, std::vector<DlCopy> &out) {
    const Dims &d = h->kp.d;
    const Job &job = *s->job;
    const size_t cap = (size_t)s->dev.cap, Nm = (size_t)d.Nm;
    if (job.dmap) {  // the 8-bit disparity images of the chunk (side 0 only): a quarter of the float maps' bytes
        if (side) return;
        uint8_t *dst8 = job.pin_out ? job.dmap + (size_t)s->i0 * Nm : s->h_out8;
        for (int j = 0; j < s->n;) {  // maximal runs of processed pairs
            if (s->h_blob[(size_t)j * META_WORDS] < 3) {
                j++;
                continue;
            }
            int e = j + 1;
            while (e < s->n && s->h_blob[(size_t)e * META_WORDS] >= 3) e++;
            out.push_back({dst8 + (size_t)j * Nm, s->d_out8 + (size_t)j * Nm, (size_t)(e - j) * Nm, false});
            j = e;
        }
        return;
    }
    float *user = side ? job.d2 : job.d1;
    if (!user) return;
    const float *dev = s->d_out + side * cap * Nm;
    float *dst = job.pin_out ? user + (size_t)s->i0 * Nm : s->h_out + side * cap * Nm;
    for (int j = 0; j < s->n;) {  // maximal runs of processed pairs
        if (s->h_blob[(size_t)j * META_WORDS] < 3) {
            j++;
            continue;
        }
        int e = j + 1;
        while (e < s->n && s->h_blob[(size_t)e * META_WORDS] >= 3) e++;
        // the runtime's copies: long runs go down as two halves on two streams (one DMA engine moved ~41 GB/s beside the uploads)
        const int half = (split && e - j >= 8) ? j + (e - j) / 2 : e;
        out.push_back({dst + (size_t)j * Nm, dev + (size_t)j * Nm, (size_t)(half - j) * Nm * sizeof(float), false});
        if (half < e) out.push_back({dst + (size_t)half * Nm, dev + (size_t)half * Nm, (size_t)(e - half) * Nm * sizeof(float), true});
        j = e;
    }
}

void download_maps(sv_handle *h, Slot *s, int side, hipStream_t st, hipStream_t st2 = nullptr) {
    std::vector<DlCopy> plan;
    plan_downloads(h, s, side, st2 != nullptr, plan);
    for (const DlCopy &c : plan) HIP_TRY(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, c.second ? st2 : st));
}

// Pageable callers: the maps of a chunk from the page-locked mirror into the caller's arrays (after ev_out)
void deliver_maps(sv_handle *h, Slot *s, int copy_helpers, int sides = 3) {  // sides: bit 0 the left maps, bit 1 the right maps
    const Dims &d = h->kp.d;
    const Job &job = *s->job;
    if (job.pin_out) return;
    const size_t cap = (size_t)s->dev.cap, Nm = (size_t)d.Nm;
    CopyList cl;
    if (s->inline_mode) cl.piece = COPY_PIECE / 4;  // a single pair's maps: enough pieces for everybody who polls
    if (job.dmap) {
        for (int j = 0; j < s->n; j++)
            if (s->h_blob[(size_t)j * META_WORDS] >= 3) cl.add(job.dmap + (size_t)(s->i0 + j) * Nm, s->h_out8 + (size_t)j * Nm, Nm);
        if (!cl.job->pieces.empty()) cl.run(h, copy_helpers);
        return;
    }
    for (int side = 0; side < 2; side++) {
        float *user = side ? job.d2 : job.d1;
        if (!user || !((sides >> side) & 1)) continue;
        for (int j = 0; j < s->n; j++)
            if (s->h_blob[(size_t)j * META_WORDS] >= 3) cl.add(user + (size_t)(s->i0 + j) * Nm, s->h_out + side * cap * Nm + (size_t)j * Nm, Nm * sizeof(float));
    }
    if (!cl.job->pieces.empty()) cl.run(h, copy_helpers);
}

void set_chunk_inputs(sv_handle *h, Slot *s) {  // device-memory jobs read the caller's tensors in place
    const Job &job = *s->job;
    s->in_pair = (size_t)h->kp.d.H * job.stride;
    s->in_left = (job.zc_left ? job.zc_left : job.left) + (size_t)s->i0 * s->in_pair;
    s->in_right = (job.zc_right ? job.zc_right : job.right) + (size_t)s->i0 * s->in_pair;
    s->in_stride = job.stride;
}

// Stage 1.  The issuer takes the chunks of the submitted batches in order, gives each the next slot of the ring, and launches its first
// GPU phase.  Host-memory batches on DMA lanes: a chunk's images have to be in HBM before its kernels are LAUNCHED (the upload engine
// and the HIP streams share no device-side dependency), so the issuer keeps the uploads of the next chunks queued on the engine -
// `ahead` chunks are prepared (slot taken, upload enqueued) before the oldest of them is waited for and launched.  The engine never
// runs dry and the host wait costs the kernels nothing: they are launched one upload behind.
void issuer_main(sv_handle *h) {
    (void)hipSetDevice(h->cfg.device);
    const int ns = (int)h->slots.size();
    std::deque<Slot *> ready;  // prepared chunks, oldest first
    Job *job = nullptr;        // the batch whose chunks are being prepared
    int next_c = 0;
    for (;;) {
        // (1) prepare: blocking while nothing is prepared, opportunistic (never waiting) beyond that
        for (;;) {
            const bool may_block = ready.empty();
            const size_t ahead = (job && next_c < job->nchunks && job->host && h->dma) ? (size_t)std::max(1, std::min(3, ns / 2)) : 1;
            if (!may_block && ready.size() >= ahead) break;
            if (!job || next_c >= job->nchunks) {
                std::unique_lock<std::mutex> lk(h->mu);
                if (may_block) h->cv.wait(lk, [&] { return h->quit || !h->jobs.empty(); });
                if (h->quit) return;
                if (h->jobs.empty()) break;
                if (!may_block && !(h->jobs.front()->host && h->dma)) break;  // (only uploads are worth preparing ahead)
                job = h->jobs.front();
                h->jobs.pop_front();
                next_c = 0;
                continue;
            }
            Slot *s = h->slots[h->ring_pos];
            bool drain = false;
            {
                const auto w0 = std::chrono::steady_clock::now();
                std::unique_lock<std::mutex> lk(h->mu);
                if (may_block) h->cv.wait(lk, [&] { return s->state != SLOT_BUSY; });
                else if (s->state == SLOT_BUSY) break;
                drain = s->state == SLOT_DRAINING;
                if (drain && !may_block && hipEventQuery(s->ev_free) != hipSuccess) break;  // its previous chunk is still in phase 2
                s->state = SLOT_BUSY;
                if (h->lat_trace) h->issue_ns[5] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
            }
            h->ring_pos = (h->ring_pos + 1) % ns;
            trace_mark(h, s, TR_SLOT);
            s->job = job;
            s->i0 = next_c * h->chunk;
            s->n = std::min(h->chunk, job->batch - s->i0);
            s->up_ticket = -1;
            next_c++;
            // who triangulates this chunk: the GPU mode keeps every chunk resident; with a balanced share every 1 / share-th chunk
            s->resident = false;
            if (h->resident_ok) {
                // Host-memory batches under the automatic policy are resident throughout: a chunk the pool triangulates moves its support
                // lists down and its blob (~120 KB per pair) up with the runtime's own copies, which take whichever SDMA engine looks free
                // when they are enqueued and then sit behind a 2 ms map download on it - phase 2 of such a chunk took 7.9 ms instead of 3
                // (tools/chunk_trace.py).  Measured with 14 pool threads / everything resident: f32 maps 22 100 - 24 600 / 27 500 pairs/s,
                // 8-bit maps 37 000 / 44 000 (profiles/HISTORY.md, round 5); the link is the limit either way, the pool has nothing to add.
                if (h->gpu_delaunay_pct >= 100 || (job->host && h->gpu_share_auto)) {
                    s->resident = true;
                } else {
                    h->issue_acc += h->gpu_share_auto ? h->shared_pct.load(std::memory_order_relaxed) : h->gpu_delaunay_pct;
                    if (h->issue_acc >= 100) {
                        h->issue_acc -= 100;
                        s->resident = true;
                    }
                }
            }
            try {
                if (drain) HIP_TRY(wait_event(h, s->ev_free));
                if (!h->failed) {
                    if (job->host) {  // images: caller -> (page-locked mirror ->) device; phase 1 waits for them
                        upload_chunk(h, s, h->sIn, 3, true);
                        trace_mark(h, s, TR_UP_QUEUED);
                        if (s->up_ticket < 0) {  // the runtime's copies on the upload stream: a device-side dependency
                            HIP_TRY(hipEventRecord(s->ev_in, h->sIn));
                            HIP_TRY(hipStreamWaitEvent(h->sP1, s->ev_in, 0));
                        }
                    } else {
                        set_chunk_inputs(h, s);
                    }
                }
            } catch (const std::exception &e) {
                note_error(h, e.what());
            }
            ready.push_back(s);
        }
        if (ready.empty()) continue;
        // (2) launch the oldest prepared chunk
        Slot *s = ready.front();
        ready.pop_front();
        g_launch_hook.fn = h->timing ? timing_hook : nullptr;
        g_launch_hook.ctx = &h->tc_issue;
        try {
            if (s->up_ticket >= 0) {
                const bool ok = h->dma->wait(s->up_ticket, h->poll_sync);
                s->up_ticket = -1;
                if (!ok) throw std::runtime_error("DMA lanes: an image upload failed");
                trace_mark(h, s, TR_UP_DONE);
            }
            if (!h->failed) issue_phase1(h, s);
            trace_mark(h, s, TR_P1_QUEUED);
        } catch (const std::exception &e) {
            note_error(h, e.what());
        }
        {
            std::lock_guard<std::mutex> lk(h->mu);
            h->q1.push_back(s);
        }
        h->cv.notify_all();
    }
}

// ---- stage 2: dispatcher + host pool ---------------------------------------------------------------------------------
void chunk_host_done(sv_handle *h, Slot *s) {
    trace_mark(h, s, TR_HOST_DONE);
    if (s->inline_mode) {
        s->inline_done.store(1, std::memory_order_release);
        return;
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->q2.push_back(s);
    }
    h->cv.notify_all();
}

void pair_done(sv_handle *h, Slot *s) {
    if (s->pending.fetch_sub(1) == 1) chunk_host_done(h, s);
}

void spawn_to_pool(void *ctx, void (*fn)(void *), void *arg) {
    sv_handle *h = static_cast<sv_handle *>(ctx);
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        Task t{nullptr, 0, 0};
        t.fn = fn;
        t.arg = arg;
        h->queue.push_front(t);
    }
    h->queue_len.fetch_add(1, std::memory_order_release);  // (after the lock is free again: the pollers go for it the moment they see this)
    if (h->pollers.load(std::memory_order_acquire) < 1) h->qcv.notify_one();  // a polling thread needs no wake-up (a system call on this thread)
}

// Latency mode: the lattice filters of the single pair as a team of the calling thread and the polling pool threads (idle until the
// filters are done: the triangulations start from their result).  The pollers watch one word of the handle beside the queue length; a
// team call opens it, whoever is polling joins and claims parts from a counter together with the caller, which does not depend on
// anybody showing up.  No lock, no allocation: a filter pass is a few microseconds and the queue's mutex would cost as much again
// (measured: ~2 us per call through the queue with seven pollers).  The caller closes the call and waits until everybody who got in
// has left before it returns - fn, arg and the counters are then free for the next call.
void team_work(sv_handle *h) {
    const int parts = h->team_parts;
    for (int q = h->team_next.fetch_add(1); q < parts; q = h->team_next.fetch_add(1)) {
        h->team_fn(h->team_arg, q);
        h->team_done.fetch_add(1);
    }
}

void team_join(sv_handle *h) {  // a polling pool thread that has seen team_open != 0
    h->team_active.fetch_add(1);
    if (h->team_open.load()) team_work(h);  // (still open after we are counted: the caller waits for us before it reuses anything)
    h->team_active.fetch_sub(1);
}

void team_run(void *ctx, int parts, void (*fn)(void *, int), void *arg) {
    sv_handle *h = static_cast<sv_handle *>(ctx);
    h->team_fn = fn, h->team_arg = arg, h->team_parts = parts;
    h->team_next.store(0);
    h->team_done.store(0);
    h->team_open.store(++h->team_calls ? h->team_calls : ++h->team_calls);
    team_work(h);
    while (h->team_done.load() < parts) __builtin_ia32_pause();
    h->team_open.store(0);
    while (h->team_active.load() != 0) __builtin_ia32_pause();
}

// one Delaunay triangulation of a pair's support points; errors are recorded, never thrown (the completion accounting of
// the caller must run in any case)
void triangulate_side(sv_handle *h, HostScratch *sc, Slot *s, int j, int side) {
    const Dims &d = h->kp.d;
    int32_t *blob = s->h_blob;
    int32_t *meta = blob + (size_t)j * META_WORDS;
    const int ns = meta[0];
    const int32_t *sup = blob + meta[1];
    if ((int)sc->xy.size() < 2 * ns) sc->xy.resize(2 * ns);
    for (int q = 0; q < ns; q++) {  // elas.cpp:449-461: left uses (u,v), right (u-d,v)
        sc->xy[2 * q] = side ? sup[3 * q] - sup[3 * q + 2] : sup[3 * q];
        sc->xy[2 * q + 1] = sup[3 * q + 1];
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (meta[7]) {  // this pair's triangulations run on the GPU
        // only the preparation stays here: [m, ids in k-d order] behind the two triangle lists; k_delaunay_blob does the rest.
        // Sets the kernel cannot hold (LDS) are triangulated here as usual and marked with m = -1.
        int32_t *ord = blob + meta[5] + (size_t)2 * ns * 3 + (size_t)side * (ns + 1);
        if (ns <= h->dg_limit) {
            const int m = sc->dl.kd_ordered_ids(sc->xy.data(), ns, ord + 1);
            if (h->timing) h->host_delaunay_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            if (m >= 0) {
                ord[0] = m;
                meta[2 + 2 * side] = 0;  // the kernel writes the count into the device copy
                return;
            }
        }
        ord[0] = -1;
        h->gpu_tri_fallbacks.fetch_add(1, std::memory_order_relaxed);
    }
    // latency mode, on request (sv_config.latency_split): the halves (2: quarters) of the top-level cuts go to other threads.  Off by
    // default: on the test hosts a KITTI set takes 136 us on one thread and 146 - 199 us shared (tools/latency_trace.py: the halves'
    // triangles come back from another core's cache for the seam and for the output pass); throughput mode keeps every core busy with
    // whole pairs anyway.  (Two triangulations at a time: halves need 4 threads, quarters 8, counting the calling thread.)
    const Delaunay::Spawn spawn{spawn_to_pool, h, (h->latency_split >= 2 && h->pool.size() >= 7) ? 2 : 1};
    const int nt = sc->dl.triangulate(sc->xy.data(), ns, blob + meta[3 + 2 * side], 2 * ns, (h->chunk == 1 && h->latency_split > 0 && h->pool.size() >= 3 && (!h->lat_auto || h->lat_near.load(std::memory_order_acquire))) ? &spawn : nullptr);
    if (h->timing) h->host_delaunay_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (nt < 0 || nt > d.max_tri) {
        note_error(h, "triangle capacity exceeded");
        meta[2 + 2 * side] = 0;
        return;
    }
    meta[2 + 2 * side] = nt;
}

// Latency mode: meta words + support points of the slot's single pair to the device and the candidate-grid kernels behind them (phase-2
// stream), while the triangulations are still being built.
void early_grid_launch(Slot *s) {
    sv_handle *h = s->grid_h;
    try {
        hipStream_t st = h->sP2[0];
        if (s->h_blob_dev && !h->lat_runtime_copies && s->grid_off * sizeof(int32_t) <= 256) {  // (one pair: the points follow the meta words)
            launch_copy_block(s->dev.blob, s->h_blob_dev, sizeof(int32_t) * (s->grid_off + (size_t)s->grid_ns * 3), st);
        } else {
            HIP_TRY(hipMemcpyAsync(s->dev.blob, s->h_blob, sizeof(int32_t) * META_WORDS, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(s->dev.blob + s->grid_off, s->h_blob + s->grid_off, sizeof(int32_t) * (size_t)s->grid_ns * 3, hipMemcpyHostToDevice, st));
        }
        launch_grid(h->kp, s->dev, 1, s->grid_ns, st);
    } catch (const std::exception &e) {
        note_error(h, e.what());
    }
}

void run_task(sv_handle *h, HostScratch *sc, const Task &t) {
    if (t.fn) {
        t.fn(t.arg);
        return;
    }
    Slot *s = t.s;
    const Dims &d = h->kp.d;
    const int lat = d.Wc * d.Hc;
    int32_t *blob = s->h_blob;
    int32_t *meta = blob + (size_t)t.pair * META_WORDS;
    if (t.side == -2) {
        // a pair of a resident chunk with a side the kernel handed back: fetch its support list and build that side the reference's way;
        // the triangle list goes to the place the device laid out for it (the host blob mirrors the device blob), phase 2 uploads it
        const int ns = meta[0];
        if ((int)sc->sup.size() < d.max_pts * 3) sc->sup.resize((size_t)d.max_pts * 3);
        if (ns > d.max_pts || hipMemcpy(sc->sup.data(), s->dev.fsup + (size_t)t.pair * d.max_pts * 3, sizeof(int32_t) * 3 * (size_t)ns, hipMemcpyDeviceToHost) != hipSuccess) {
            note_error(h, "hipMemcpy of a support list failed");
            meta[0] = 0;
        } else {
            if ((int)sc->xy.size() < 2 * ns) sc->xy.resize(2 * ns);
            for (int side = 0; side < 2; side++) {
                if (meta[2 + 2 * side] >= 0) continue;
                for (int q = 0; q < ns; q++) {  // elas.cpp:449-461: left uses (u,v), right (u-d,v)
                    sc->xy[2 * q] = side ? sc->sup[3 * q] - sc->sup[3 * q + 2] : sc->sup[3 * q];
                    sc->xy[2 * q + 1] = sc->sup[3 * q + 1];
                }
                const auto t0 = std::chrono::steady_clock::now();
                const int nt = sc->dl.triangulate(sc->xy.data(), ns, blob + meta[3 + 2 * side], 2 * ns);
                if (h->timing) h->host_delaunay_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                if (nt < 0 || nt > d.max_tri) {
                    note_error(h, "triangle capacity exceeded");
                    meta[2 + 2 * side] = 0;
                } else {
                    meta[2 + 2 * side] = nt;
                }
                meta[6] |= 1 << side;
                h->gpu_tri_fallbacks.fetch_add(1, std::memory_order_relaxed);
            }
        }
        pair_done(h, s);
        return;
    }
    if (t.side >= 0) {
        triangulate_side(h, sc, s, t.pair, t.side);
        if (__atomic_add_fetch(&meta[6], 1, __ATOMIC_ACQ_REL) == 2) pair_done(h, s);
        return;
    }
    if ((int)sc->sup.size() < d.max_pts * 3) sc->sup.resize((size_t)d.max_pts * 3);
    const auto t_task0 = std::chrono::steady_clock::now();
    int ns;
    if (h->gpu_filter) {  // the lattice filters already ran on the GPU (k_support_filter): just pick up the list
        ns = s->h_fnsup[t.pair];
        if (ns < 0 || ns > d.max_pts) {
            note_error(h, "corrupt support count from the GPU filter");
            ns = 0;
        }
        if (ns <= fsup_copy_pts(d)) {
            memcpy(sc->sup.data(), s->h_fsup + (size_t)t.pair * fsup_copy_pts(d) * 3, sizeof(int32_t) * 3 * (size_t)ns);
        } else if (hipMemcpy(sc->sup.data(), s->dev.fsup + (size_t)t.pair * d.max_pts * 3, sizeof(int32_t) * 3 * (size_t)ns, hipMemcpyDeviceToHost) != hipSuccess) {
            note_error(h, "hipMemcpy of a long support list failed");
            ns = 0;
        }
    } else {
        const auto tf0 = std::chrono::steady_clock::now();
        // work on a private copy: the filters rewrite the lattice in place and the pinned buffer is DMA-visible memory
        const FilterTeam team{team_run, h, 1 + h->pollers.load(std::memory_order_acquire)};
        const bool shared = s->inline_mode && h->latency_split > 0 && h->lat_pin && h->lat_near.load(std::memory_order_acquire) && !h->lat_filter_alone && team.threads > 1 && support_filter_team_usable(h->p);
        if (shared) {
            // the pollers sit next to this thread: the filters as a team.  The team reads the lattice where the device left it (its first
            // pass only reads, every thread its own columns) and writes the result over it at the end; the slot has one pair, so the
            // sixteen entries the vector loads may touch behind the lattice are the buffer's padding.
            ns = support_filter_t(h->p, s->h_dcan + (size_t)t.pair * lat, d.W, d.H, sc->sup.data(), d.max_pts, &team, &sc->filter);
        } else {
            if ((int)sc->lattice.size() < lat + LATTICE_PAD) sc->lattice.assign((size_t)lat + LATTICE_PAD, 0);
            memcpy(sc->lattice.data(), s->h_dcan + (size_t)t.pair * lat, sizeof(int16_t) * (size_t)lat);
            ns = support_filter_t(h->p, sc->lattice.data(), d.W, d.H, sc->sup.data(), d.max_pts);
        }
        if (h->timing) h->host_filter_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tf0).count();
        if (ns < 0) {
            note_error(h, "support point capacity exceeded");
            ns = 0;
        }
    }
    if (h->timing) h->host_tasks += 1;
    const auto t_filtered = std::chrono::steady_clock::now();
    if (s->job->status) s->job->status[s->i0 + t.pair] = ns;
    meta[0] = ns;
    meta[1] = meta[3] = meta[5] = 0;
    meta[2] = meta[4] = 0;
    meta[6] = 0;
    {  // which pairs are triangulated on the GPU: every pair in the GPU mode, an evenly spread share of them otherwise
        const long idx = (long)s->i0 + t.pair, pct = s->gpu_pct;
        meta[7] = (pct >= 100 || ((idx + 1) * pct / 100 != idx * pct / 100)) && pct > 0 ? 1 : 0;
        h->tri_pairs.fetch_add(1, std::memory_order_relaxed);
        if (meta[7]) h->gpu_tri_pairs.fetch_add(1, std::memory_order_relaxed);
    }
    if (ns < 3) {  // elas.cpp:63-69
        pair_done(h, s);
        return;
    }
    // 3*ns words of points + two triangle lists of at most 2*ns triangles each
    const size_t need = (size_t)ns * 3 + 2 * ((size_t)2 * ns * 3) + (meta[7] ? 2 * ((size_t)ns + 1) : 0);
    const size_t off = s->blob_off.fetch_add(need);
    if (off + need > s->blob_words) {
        note_error(h, "host blob overflow");
        meta[0] = 0;
        pair_done(h, s);
        return;
    }
    meta[1] = (int32_t)off;
    meta[3] = (int32_t)(off + (size_t)ns * 3);
    meta[5] = (int32_t)(off + (size_t)ns * 3 + (size_t)2 * ns * 3);
    memcpy(blob + off, sc->sup.data(), sizeof(int32_t) * (size_t)ns * 3);
    // latency mode (this is the calling thread): the candidate grid only needs the support points - they are uploaded and the grid is
    // launched now, so that it runs while the two triangulations are built.  With two or more pool threads one of them makes those four
    // runtime calls (15 - 20 us) while this thread starts on the left triangulation; the chunk counts it as one more pending piece.
    const bool early_grid = s->inline_mode && !meta[7];
    const bool grid_on_pool = early_grid && h->pool.size() >= 2;
    if (early_grid) {
        s->grid_h = h, s->grid_off = off, s->grid_ns = ns;
        s->grid_issued = true;  // (phase 2 is enqueued only after every pending piece has reported)
        if (grid_on_pool)
            s->pending.fetch_add(1);
        else
            early_grid_launch(s);
    }
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        if (grid_on_pool) {
            Task g{nullptr, 0, 0};
            g.fn = [](void *arg) {
                Slot *sl = static_cast<Slot *>(arg);
                early_grid_launch(sl);
                pair_done(sl->grid_h, sl);
            };
            g.arg = s;
            h->queue.push_front(g);
        }
        h->queue.push_front(Task{s, t.pair, 1});
    }
    h->queue_len.fetch_add(grid_on_pool ? 2 : 1, std::memory_order_release);  // published once, with the lock free again: pollers go for it at once
    // (a poller that gives up just now still finds the queue non-empty when it takes the lock to wait: nothing is lost without a notify)
    if (h->pollers.load(std::memory_order_acquire) < (grid_on_pool ? 2 : 1)) {
        if (grid_on_pool)
            h->qcv.notify_all();
        else
            h->qcv.notify_one();
    }
    const auto t_handed = std::chrono::steady_clock::now();
    triangulate_side(h, sc, s, t.pair, 0);
    if (s->inline_mode && h->lat_trace) {
        const auto t_end = std::chrono::steady_clock::now();
        h->lat_sub_ns[0] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t_filtered - t_task0).count();
        h->lat_sub_ns[1] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t_handed - t_filtered).count();
        h->lat_sub_ns[2] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t_end - t_handed).count();
    }
    if (__atomic_add_fetch(&meta[6], 1, __ATOMIC_ACQ_REL) == 2) pair_done(h, s);
}

// pool threads that poll for a frame's pieces in latency mode: the right triangulation and the early grid launch, plus the halves
// (quarters) of the two triangulations when those are shared
int latency_pollers(const sv_handle *h) { return 2 + (h->latency_split >= 2 ? 5 : (h->latency_split == 1 ? 2 : 0)); }  // (quarters: 1 + 1 + 3 + 3 pieces - the grid launch is over before the quarters are cut, and an L3 domain of 8 cores has 7 beside the caller's)

void pool_main(sv_handle *h, HostScratch *sc, int idx) {
    (void)hipSetDevice(h->cfg.device);  // run_task may fetch a long support list from the handle's device
    // Latency mode (chunk 1): a frame hands over the right triangulation and the early grid launch (and, on request, pieces of the
    // triangulations) within ~0.2 ms, and waking a thread that sleeps on the condition variable costs 30-50 us each time.  There the
    // first few pool threads - as many as a frame has pieces - poll the queue length for a while (about a frame period of continuous
    // use) before they go to sleep; the others, and throughput handles, sleep at once.  (All 14 polling: 15 busy threads on a 16-CPU
    // quota, and a notify_all that costs the calling thread 19 us.)  A producer that finds enough pollers does not notify at all.
    const int hot = latency_pollers(h);
    int team_seen = 0;
    for (;;) {
        Task t;
        bool have = false;
        const int spin_rounds = h->chunk == 1 && idx < hot && !h->pool_sleep ? 400000 : 0;
        if (spin_rounds) {
            // a poller never sleeps on the queue's mutex (try_lock): nobody then has to wake it with a system call when the lock is released
            h->pollers.fetch_add(1, std::memory_order_acq_rel);
            for (int i = 0; i < spin_rounds && !have; i++) {
                if (const int call = h->team_open.load(std::memory_order_acquire); call && call != team_seen) {
                    team_join(h);
                    team_seen = call;  // (once per call: no parts are left when team_join returns)
                    i = 0;             // in use: keep polling
                }
                if (h->queue_len.load(std::memory_order_acquire) > 0 && h->qmu.try_lock()) {
                    if (!h->queue.empty()) {
                        t = h->queue.front();
                        h->queue.pop_front();
                        h->queue_len.fetch_sub(1, std::memory_order_relaxed);
                        have = true;
                    }
                    h->qmu.unlock();
                }
                if (!have) __builtin_ia32_pause();
            }
            h->pollers.fetch_sub(1, std::memory_order_acq_rel);
        }
        if (!have) {
            std::unique_lock<std::mutex> lk(h->qmu);
            const auto ready = [&] { return h->pool_quit || !h->queue.empty(); };
            bool timed = false;
            if (spin_rounds) {
                // A poller that has run out of patience.  Frames at a regular pace (a camera at 30 Hz: one call every 33 ms, far beyond the
                // polling window): sleep towards the next one and be polling again shortly before it is due - a frame that finds the pollers
                // asleep pays 30 - 50 us per wake-up and runs its host stage mostly alone (measured at one frame per 33 ms: 0.51 ms instead of 0.28).
                using clk = std::chrono::steady_clock;
                const int64_t next = h->lat_next_expected_ns.load(std::memory_order_relaxed), period = next ? h->lat_period_ns_pub.load(std::memory_order_relaxed) : 0;
                const int64_t lead = h->lat_wake_lead_ns > 0 ? h->lat_wake_lead_ns : std::max<int64_t>(300000, std::min<int64_t>(2000000, period / 16));
                const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now().time_since_epoch()).count();
                if (next - lead > now + 100000) {
                    timed = true;
                    if (!h->qcv.wait_until(lk, clk::time_point(std::chrono::duration_cast<clk::duration>(std::chrono::nanoseconds(next - lead))), ready)) continue;  // time to poll again
                }
            }
            if (!timed) h->qcv.wait(lk, ready);
            if (h->pool_quit && h->queue.empty()) return;
            t = h->queue.front();
            h->queue.pop_front();
            h->queue_len.fetch_sub(1, std::memory_order_relaxed);
        }
        run_task(h, sc, t);
    }
}

void dispatcher_main(sv_handle *h) {
    (void)hipSetDevice(h->cfg.device);
    for (;;) {
        Slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return h->quit || !h->q1.empty(); });
            if (h->quit) return;
            s = h->q1.front();
            h->q1.pop_front();
        }
        bool ok = !h->failed;
        if (ok) {
            if (wait_event(h, s->ev_p1) != hipSuccess) {
                note_error(h, "hipEventSynchronize(phase 1) failed");
                ok = false;
            }
        }
        trace_mark(h, s, TR_P1_DONE);
        if (!ok) {  // keep the pipeline moving: mark every pair as skipped
            for (int j = 0; j < s->n; j++) {
                int32_t *meta = s->h_blob + (size_t)j * META_WORDS;
                for (int q = 0; q < META_WORDS; q++) meta[q] = 0;
            }
            s->blob_off.store((size_t)s->dev.cap * META_WORDS);
            chunk_host_done(h, s);
            continue;
        }
        if (h->gpu_share_auto) {  // (the balance itself: see below; with resident chunks the issuer applies it, chunk by chunk)
            const int backlog = h->queue_len.load(std::memory_order_acquire);
            if (backlog > s->n) h->auto_pct = std::min(h->auto_pct + 2, 95);
            else if (backlog * 4 <= s->n) h->auto_pct = std::max(h->auto_pct - 2, 0);
            h->shared_pct.store(h->auto_pct, std::memory_order_relaxed);
        }
        if (s->resident) {
            // The device built this chunk's blob: the meta words are here (h_blob).  What is left for the host: the callers' status, the
            // size the next launches' LDS request follows, and the sides the kernel handed back (triangle count -1: coincident points
            // - the reference's quicksort decides which of them survives -, or more vertices than the launch had LDS for).
            int nfb = 0, seen = 0;
            for (int j = 0; j < s->n; j++) {
                int32_t *meta = s->h_blob + (size_t)j * META_WORDS;
                const int ns = meta[0];
                if (s->job->status) s->job->status[s->i0 + j] = ns;
                meta[6] = 0;  // host copy: which sides the host stage builds (bit 0 left, bit 1 right)
                if (ns <= h->dg_sub_max) seen = std::max(seen, ns);  // (larger sets take the cut path: they do not size the LDS request)
                if (ns >= 3 && (meta[2] < 0 || meta[4] < 0)) nfb++;
            }
            h->tri_pairs.fetch_add(s->n, std::memory_order_relaxed);
            h->gpu_tri_pairs.fetch_add(s->n, std::memory_order_relaxed);
            {  // grow at once, shrink slowly: a launch whose LDS request is too small hands its large sets to the host
                const int target = std::min(h->dg_sub_max, seen + seen * h->ns_margin_pct / 100 + 32), cur = h->ns_bound.load(std::memory_order_relaxed);
                h->ns_bound.store(target > cur ? target : cur - (cur - target + 3) / 4, std::memory_order_relaxed);
            }
            s->gpu_pct = 0;
            if (nfb == 0) {
                chunk_host_done(h, s);
                continue;
            }
            s->pending.store(nfb);
            {
                std::lock_guard<std::mutex> lk(h->qmu);
                for (int j = 0; j < s->n; j++) {
                    const int32_t *meta = s->h_blob + (size_t)j * META_WORDS;
                    if (meta[0] >= 3 && (meta[2] < 0 || meta[4] < 0)) {
                        h->queue.push_back(Task{s, j, -2});
                        h->queue_len.fetch_add(1, std::memory_order_release);
                    }
                }
            }
            h->qcv.notify_all();
            continue;
        }
        s->blob_off.store((size_t)s->dev.cap * META_WORDS);
        s->pending.store(s->n);
        s->gpu_pct = h->resident_ok ? 0 : h->gpu_delaunay_pct;
        if (h->gpu_share_auto && !h->resident_ok) {
            // Tasks of earlier chunks that no pool thread has picked up yet when the next lattice arrives: the pool is behind the GPU,
            // so the triangulation kernel takes a larger share of this chunk; a short queue gives the share back to the pool.  (One
            // chunk at a time - a single slot, a profile's serial pass - never finds a backlog and stays on the host.)
            // (thresholds from sweeps on one MI355X: more than one chunk of unstarted tasks: up 2 points, less than a quarter: down 2 -
            //  steps of 5 at 1.5 / 0.5 chunks gave the same mean rate with more scatter; settles at 13-30 % with 14 threads, 26 % with 12,
            //  47 % with 10, 60 % with 8, 74 % with 6, 86 % with 4 - each 1-2 % above what the all-GPU mode reaches)
            // The share is handed out in whole chunks (every 1 / share-th chunk goes to the GPU kernel entirely) rather than as a
            // slice of every chunk: the kernel is a latency chain of ~0.9 ms whatever the number of sets (one workgroup each), so
            // fewer, fuller launches cost the same GPU time per set and a quarter of the launches.  SV_GPU_DELAUNAY_SLICED=1: the old way.
            if (h->share_sliced) {
                s->gpu_pct = h->auto_pct;
            } else {
                h->auto_acc += h->auto_pct;
                s->gpu_pct = 0;
                if (h->auto_acc >= 100) {
                    h->auto_acc -= 100;
                    s->gpu_pct = 100;
                }
            }
        }
        {
            std::lock_guard<std::mutex> lk(h->qmu);
            for (int j = 0; j < s->n; j++) h->queue.push_back(Task{s, j, -1});
            h->queue_len.fetch_add(s->n, std::memory_order_release);
        }
        h->qcv.notify_all();
    }
}

// ---- stage 3: finisher -------------------------------------------------------------------------------------------------
void issue_phase2(sv_handle *h, Slot *s, hipStream_t st) {
    const KParams &k = h->kp, &km = h->kp_map;  // image-space / map-space parameters
    const Dims &d = k.d;
    const Job &job = *s->job;
    const bool dbg = h->cfg.keep_debug != 0;
    const int n = s->n;
    int32_t *blob = s->h_blob;
    const size_t off = s->blob_off.load();
    if (dbg) {
        const int32_t *meta = blob + (size_t)(n - 1) * META_WORDS;
        dbg_put(h, "support", blob + meta[1], meta[0] >= 3 ? (size_t)meta[0] * 3 : 0);
        dbg_put(h, "tri1", blob + meta[3], (size_t)meta[2] * 3);
        dbg_put(h, "tri2", blob + meta[5], (size_t)meta[4] * 3);
    }
    if (s->resident) {  // the device has the blob already; only what the host stage built for handed-back sides goes up
        for (int j = 0; j < n; j++) {
            int32_t *meta = blob + (size_t)j * META_WORDS;
            const int mask = meta[6];
            if (!mask) continue;
            meta[6] = 0;
            for (int side = 0; side < 2; side++)
                if ((mask >> side) & 1 && meta[2 + 2 * side] > 0)
                    HIP_TRY(hipMemcpyAsync(s->dev.blob + meta[3 + 2 * side], blob + meta[3 + 2 * side], sizeof(int32_t) * 3 * (size_t)meta[2 + 2 * side], hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(s->dev.blob + (size_t)j * META_WORDS, meta, sizeof(int32_t) * META_WORDS, hipMemcpyHostToDevice, st));
        }
    } else if (s->inline_mode && s->h_blob_dev && !h->lat_runtime_copies) {
        launch_copy_block(s->dev.blob, s->h_blob_dev, off * sizeof(int32_t), st);  // single pair: ~0.13 MB in stream order, by a kernel
    } else {
        HIP_TRY(hipMemcpyAsync(s->dev.blob, blob, off * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    if (!s->resident && s->gpu_pct > 0) {  // triangle lists still missing: built on the device from the vertex orders the host left in the blob
        int ns_max = 0, ns_large = 0;  // largest flagged set the LDS kernel takes / the cut path takes (larger ones: the host did them)
        for (int j = 0; j < n; j++) {
            const int ns = blob[(size_t)j * META_WORDS];
            if (!blob[(size_t)j * META_WORDS + 7]) continue;
            if (ns <= h->dg_sub_max) ns_max = std::max(ns_max, ns);
            else if (ns <= h->dg_limit) ns_large = std::max(ns_large, ns);
        }
        if (ns_max >= 3) launch_delaunay_blob(s->dev.blob, n, delaunay_gpu_lds_bytes(ns_max, ns_max), h->dg_sub_max, st);
        if (ns_large > 0) launch_delaunay_blob_large(s->dev.blob, n, ns_large, h->dg_sub_max, s->dev.dg, st);
    }
    int max_points = 0;  // the chunk's largest support-point count: sizes the grids of the per-point / per-triangle kernels
    for (int j = 0; j < n; j++) max_points = std::max(max_points, blob[(size_t)j * META_WORDS]);
    if (!s->grid_issued) launch_grid(k, s->dev, n, max_points, st);  // (latency mode launches it during the triangulations, run_task)
    s->grid_issued = false;
    launch_triangles(k, s->dev, n, max_points, st);
    launch_dense(k, s->dev, n, st);
    float *u1 = job.d1 ? job.d1 + (size_t)s->i0 * d.Nm : nullptr, *u2 = job.d2 ? job.d2 + (size_t)s->i0 * d.Nm : nullptr;  // the caller's maps are [batch][Hm][Wm]
    if (job.host && job.zc_d1) {  // a single pair, page-locked maps: the kernels write them over PCIe themselves
        u1 = job.zc_d1;
        u2 = job.zc_d2;
        // ... except a right map that is final after the L/R check (postprocess_only_left): the rest of phase 2 would wait for its
        // 1.86 MB to cross the link - it goes to the staging buffer and a DMA engine takes it down meanwhile (run_inline, ev_lr)
        if (h->nproc == 1 && job.d2) u2 = s->d_out + (size_t)s->dev.cap * d.Nm;
    } else if (job.host) {  // host-memory job: the maps are written to the slot's device staging and downloaded from there
        u1 = s->d_out;
        u2 = job.d2 ? s->d_out + (size_t)s->dev.cap * d.Nm : nullptr;
    }
    const bool only_left = h->nproc == 1;
    // with postprocess_only_left the checked right map is final: it goes straight to the caller (or nowhere)
    launch_lr(km, s->dev, n, st, only_left ? u2 : nullptr, !only_left || dbg);
    if (job.host && only_left && u2) HIP_TRY(hipEventRecord(s->ev_lr, st));  // ... and its download may overlap the rest of phase 2
    const bool active = dbg && blob[(size_t)(n - 1) * META_WORDS] >= 3;
    if (active) {
        const int j = n - 1;
        const int32_t *meta = blob + (size_t)j * META_WORDS;
        dbg_grid(h, st, s, j);
        dbg_from_device(h, st, "planes1", s->dev.planes + ((size_t)j * 2) * d.max_tri * 6, (size_t)meta[2] * 6 * sizeof(float));
        dbg_from_device(h, st, "planes2", s->dev.planes + ((size_t)j * 2 + 1) * d.max_tri * 6, (size_t)meta[4] * 6 * sizeof(float));
        dbg_from_device(h, st, "tri_id1", s->dev.tri_id + ((size_t)j * 2) * d.N, (size_t)d.N * 4);
        dbg_from_device(h, st, "tri_id2", s->dev.tri_id + ((size_t)j * 2 + 1) * d.N, (size_t)d.N * 4);
        dbg_maps_i16(h, st, "wta", s->dev.wta, j);
        dbg_maps(h, st, "lr", s->dev.disp, j);
    }
    launch_speckle(km, s->dev, n, h->nproc, st);
    if (active) dbg_maps(h, st, "speckle", s->dev.disp, n - 1);
    launch_gap_rows(km, s->dev, n, h->nproc, st);
    launch_gap_cols(km, s->dev, n, h->nproc, st);
    if (active) dbg_maps(h, st, "gap", s->dev.disp, n - 1);
    // the two separable filters are fused, out-of-place kernels: the maps ping-pong between `disp` and `tmp`
    float *cur = s->dev.disp, *alt = s->dev.tmp;
    if (h->p.filter_adaptive_mean) {
        launch_amean(km, s->dev, n, h->nproc, st, cur, alt);
        std::swap(cur, alt);
    }
    if (active) dbg_maps_nproc(h, st, "amean", cur, s->dev.disp, n - 1);
    if (h->p.filter_median) {  // the last stage writes the caller's maps itself
        launch_median(km, s->dev, n, h->nproc, st, cur, dbg ? alt : nullptr, u1, only_left ? nullptr : u2);
        std::swap(cur, alt);
    } else {
        launch_output(km, s->dev, n, cur, u1, only_left ? nullptr : u2, st);
    }
    if (active) dbg_maps_nproc(h, st, "final", cur, s->dev.disp, n - 1);
    if (job.host && job.dmap && launch_disp_to_u8(s->d_out, (size_t)n * d.Nm, s->d_out8, st) != 0)  // leftdpf.convertTo(dmap, CV_8UC1, 4.0), stereo_vision.cpp:316
        throw std::runtime_error("disparity to 8-bit conversion failed to launch");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_free, st));
    if (job.host) HIP_TRY(hipEventRecord(s->ev_p2, st));
}

// Host-memory jobs: the maps of a chunk from the device staging to the caller (or the page-locked mirror), after issue_phase2.
// Issued by the drainer thread (the calling thread in latency mode), NOT by the finisher: hipMemcpyAsync may hold its caller
// until the copy has been handed to a DMA engine, and the finisher has the next chunk's kernels to enqueue meanwhile.
void download_chunk(sv_handle *h, Slot *s) {
    const Job &job = *s->job;
    const bool only_left = h->nproc == 1;
    if (only_left && job.d2) {  // final since the L/R check: usually long done when the drainer gets here
        HIP_TRY(hipStreamWaitEvent(h->sOut, s->ev_lr, 0));
        download_maps(h, s, 1, h->sOut);
    }
    HIP_TRY(hipStreamWaitEvent(h->sOut, s->ev_p2, 0));
    HIP_TRY(hipStreamWaitEvent(h->sOut2, s->ev_p2, 0));
    download_maps(h, s, 0, h->sOut, h->sOut2);
    if (!only_left) download_maps(h, s, 1, h->sOut, h->sOut2);
    HIP_TRY(hipEventRecord(s->ev_out, h->sOut));
    HIP_TRY(hipStreamSynchronize(h->sOut2));
    HIP_TRY(wait_event(h, s->ev_out));
}

// The same on DMA lanes: the host waits for phase 2 (nothing sits in an engine's queue behind a wait), the copies go to the download
// engines - left maps / 8-bit images on one, right maps on the other - and s->down_ticket names them: the deliverer waits for the
// ticket, so the drainer can queue the next chunk's copies while these run.
void download_chunk_lanes(sv_handle *h, Slot *s) {
    const Job &job = *s->job;
    s->down_ticket = -1;
    HIP_TRY(wait_event(h, s->ev_p2));
    trace_mark(h, s, TR_P2_DONE);
    std::vector<DlCopy> plan;
    size_t n1 = 0;
    if (job.d2 && !job.dmap) plan_downloads(h, s, 1, false, plan);
    n1 = plan.size();
    plan_downloads(h, s, 0, false, plan);
    if (plan.empty()) return;
    s->down_ticket = h->dma->begin((int)plan.size());
    if (s->down_ticket < 0) throw std::runtime_error("DMA lanes: no completion signal");
    bool ok = true;
    for (size_t i = 0; i < plan.size(); i++)
        ok = h->dma->add(s->down_ticket, i < n1 ? DmaLanes::DOWN2 : DmaLanes::DOWN, plan[i].dst, plan[i].src, plan[i].bytes, false) && ok;
    if (!ok) note_error(h, "DMA lanes: a map download was refused");
    trace_mark(h, s, TR_DOWN_QUEUED);
}

void finisher_main(sv_handle *h) {
    (void)hipSetDevice(h->cfg.device);
    size_t rr = 0;
    for (;;) {
        Slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return h->quit || !h->q2.empty(); });
            if (h->quit) return;
            s = h->q2.front();
            h->q2.pop_front();
        }
        g_launch_hook.fn = h->timing ? timing_hook : nullptr;
        g_launch_hook.ctx = &h->tc_finish;
        bool recorded = false;
        try {
            if (!h->failed) {
                hipStream_t st = h->sP2[rr++ % h->sP2.size()];
                issue_phase2(h, s, st);
                trace_mark(h, s, TR_P2_QUEUED);
                recorded = true;
                const size_t c = (size_t)(s->i0 / h->chunk);
                if (!s->job->host && c < s->job->ev_done.size() && s->job->ev_done[c]) HIP_TRY(hipEventRecord(s->job->ev_done[c], st));
            }
        } catch (const std::exception &e) {
            note_error(h, e.what());
        }
        {
            std::lock_guard<std::mutex> lk(h->mu);
            if (s->job->host) {  // the slot stays busy until the drainer has seen its maps arrive
                s->job->issued2++;
                s->out_enqueued = recorded;
                h->q3.push_back(s);
            } else {
                s->state = recorded ? SLOT_DRAINING : SLOT_FREE;
                if (++s->job->issued2 == s->job->nchunks) h->jobs_finished++;
            }
        }
        h->cv.notify_all();
    }
}

// ---- stage 4 (host-memory jobs only): drainer ---------------------------------------------------------------------------
// Waits for a chunk's map downloads, hands pageable callers their maps, frees the slot and counts the job's chunks.
void drainer_main(sv_handle *h) {
    (void)hipSetDevice(h->cfg.device);
    for (;;) {
        Slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return h->quit || !h->q3.empty(); });
            if (h->quit) return;
            s = h->q3.front();
            h->q3.pop_front();
        }
        s->delivered_ok = false;
        s->down_ticket = -1;
        if (s->out_enqueued && !h->failed) {
            try {
                const auto t0 = std::chrono::steady_clock::now();
                if (h->lat_trace) (void)hipEventSynchronize(s->ev_p2);  // trace only: separates "waiting for phase 2" from the copy
                const auto t1 = std::chrono::steady_clock::now();
                if (h->dma) download_chunk_lanes(h, s);
                else download_chunk(h, s);
                s->delivered_ok = true;
                if (h->lat_trace) {
                    h->drain_ns[0] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
                    h->drain_ns[1] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
                    h->drain_chunks++;
                }
            } catch (const std::exception &e) {
                note_error(h, e.what());
            }
        }
        {  // the maps are in host memory (the caller's, or the page-locked mirror): the deliverer takes it from here, so that the next
           // chunk's download runs while a pageable caller's maps are being copied over
            std::lock_guard<std::mutex> lk(h->mu);
            h->q4.push_back(s);
        }
        h->cv.notify_all();
    }
}

// ---- stage 5 (host-memory jobs only): deliverer ---------------------------------------------------------------------------
// Pageable callers: page-locked mirror -> the caller's arrays.  Frees the slot and counts the job's chunks.
void deliverer_main(sv_handle *h) {
    for (;;) {
        Slot *s = nullptr;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv.wait(lk, [&] { return h->quit || !h->q4.empty(); });
            if (h->quit) return;
            s = h->q4.front();
            h->q4.pop_front();
        }
        if (s->down_ticket >= 0) {  // DMA lanes: the chunk's downloads are in flight (or queued behind the previous chunk's)
            if (!h->dma->wait(s->down_ticket, h->poll_sync)) {
                note_error(h, "DMA lanes: a map download failed");
                s->delivered_ok = false;
            }
            s->down_ticket = -1;
        }
        trace_mark(h, s, TR_DOWN_DONE);
        if (s->delivered_ok && !h->failed) {
            const auto t0 = std::chrono::steady_clock::now();
            deliver_maps(h, s, 3);
            if (h->lat_trace) h->drain_ns[2] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        }
        {
            std::lock_guard<std::mutex> lk(h->mu);
            s->state = SLOT_FREE;
            trace_mark(h, s, TR_FREE);
            if (++s->job->drained == s->job->nchunks) h->jobs_finished++;
        }
        h->cv.notify_all();
    }
}

// Host cores this process may use: the cgroup CPU quota when there is one, else the affinity mask; shared evenly between
// the ranks of one node (LOCAL_WORLD_SIZE, set by torch.distributed.run).  Without a visible quota the pool stays at 16.
// cores this rank may use (cgroup quota or affinity mask, divided by the ranks of the node); *quota = a cgroup quota is set
double host_cpu_share(bool *quota, bool ignore_quota = false) {
    double cpus = 0.0;
    if (FILE *f = ignore_quota ? nullptr : fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64];
        double period = 0.0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) cpus = atof(q) / period;
        fclose(f);
    }
    if (cpus <= 0.0 && !ignore_quota) {  // cgroup v1
        double quota = -1.0, period = 0.0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f, "%lf", &quota) != 1) quota = -1.0;
            fclose(f);
        }
        if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(f, "%lf", &period) != 1) period = 0.0;
            fclose(f);
        }
        if (quota > 0.0 && period > 0.0) cpus = quota / period;
    }
    const bool have_quota = cpus > 0.0;
    cpu_set_t set;
    CPU_ZERO(&set);
    int aff = (int)std::max(1u, std::thread::hardware_concurrency());
    if (sched_getaffinity(0, sizeof(set), &set) == 0) aff = std::max(1, CPU_COUNT(&set));
    int ranks = 1;
    if (env_int("LOCAL_WORLD_SIZE", &ranks)) ranks = std::max(1, ranks);
    if (quota) *quota = have_quota;
    // a quota is the node's budget and is shared between the node's ranks; an affinity mask narrower than that share is the rank's own
    // already (a launcher that pins its ranks) and is not divided again
    if (have_quota) return std::min((double)aff, std::max(1.0, cpus / ranks));
    // no quota: a mask that is already no wider than one rank's share of the machine (launcher.restrict_to_host_share, torchrun ranks
    // pinned per rank) is not divided a second time - the pool would be sized for 1 / ranks^2 of the CPUs
    const long hw = std::max(1L, sysconf(_SC_NPROCESSORS_CONF));  // the machine's CPUs, whatever this process's mask
    if (ranks > 1 && (long)aff * ranks <= hw) return (double)aff;
    return (double)aff / ranks;
}

int default_pool_size(bool ignore_quota = false) {
    bool have_quota = false;
    const int share = std::max(1, (int)host_cpu_share(&have_quota, ignore_quota));
    // under a CPU quota the pool leaves two cores to the control threads and the caller: a pool that fills the quota gets the whole
    // process throttled in bursts (sustained over 20 000 pairs: 37 300 - 38 600 pairs/s with 16 threads of a 16-CPU quota,
    // 40 100 - 40 400 with 14)
    return std::max(1, std::min(have_quota ? 32 : 16, have_quota && share > 4 ? share - 2 : share));
}

// CPUs of the NUMA node the GPU hangs off (its PCI device's numa_node), within this process's affinity mask.  The handle's threads
// are bound to them: their page-locked buffers live on that node (the runtime places hipHostMalloc memory next to the device),
// and a process with a CPU quota but the whole machine in its mask (16 of 256 CPUs on the test boxes) is otherwise moved between
// the sockets.  Empty (no binding) when the topology cannot be read, the node has no allowed CPU, or SV_NO_AFFINITY is set.
bool gpu_node_cpus(int device, cpu_set_t *out) {
    CPU_ZERO(out);
    char bdf[64] = {0}, path[256], buf[4096];
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) return false;
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    int n = 0;
    for (char *p = buf; *p;) {  // "0-63,128-191"
        char *end;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, out);
                n++;
            }
        while (*p == ',' || *p == '\n' || *p == ' ') p++;
    }
    return n > 0;
}

void bind_thread(std::thread &t, const cpu_set_t *set) {
    if (set) (void)pthread_setaffinity_np(t.native_handle(), sizeof(cpu_set_t), set);
}

void name_thread(std::thread &t, const char *name) { (void)pthread_setname_np(t.native_handle(), name); }  // (shows in /proc/<pid>/task/*/comm: tools/thread_cpu.py)

// "0-7,128-135" -> CPUs
static void parse_cpu_list(const char *buf, cpu_set_t *out) {
    CPU_ZERO(out);
    for (const char *p = buf; *p;) {
        char *end;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0) CPU_SET((int)c, out);
        while (*p == ',' || *p == '\n' || *p == ' ') p++;
    }
}

static bool read_cpu_list(const char *fmt, int cpu, cpu_set_t *out) {
    char path[160], buf[1024];
    snprintf(path, sizeof(path), fmt, cpu);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!got) return false;
    parse_cpu_list(buf, out);
    return CPU_COUNT(out) > 0;
}

// Latency mode with shared triangulations: the pieces a frame hands to pool threads come back through the caches (the halves' triangles
// are read by the seam merge and the output pass of the calling thread).  From another core complex that costs more than the second
// thread saves (EPYC 9575F, a KITTI set: 134 us on one thread, 181 - 195 us in halves on threads anywhere on the socket, 106 - 112 us
// with the helper on a core that shares the caller's L3), so the polling pool threads are kept on the calling thread's L3 domain.
//
// One CPU per core of `cpu`'s L3 domain within the process's mask, the core of `cpu` itself left out; l3_first = the domain's first CPU.
static bool l3_helper_cores(int cpu, int want, std::vector<int> *cores, int *l3_first) {
    cpu_set_t l3, allowed, taken;
    if (cpu < 0 || !read_cpu_list("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu, &l3)) return false;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    if (!read_cpu_list("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu, &taken)) {
        CPU_ZERO(&taken);
        CPU_SET(cpu, &taken);
    }
    *l3_first = -1;
    cores->clear();
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &l3)) continue;
        if (*l3_first < 0) *l3_first = c;
        if (!CPU_ISSET(c, &allowed) || CPU_ISSET(c, &taken)) continue;
        cores->push_back(c);
        cpu_set_t sib;
        if (read_cpu_list("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c, &sib)) CPU_OR(&taken, &taken, &sib);
    }
    return (int)cores->size() >= want;
}

// Called at the start of every single-pair call: re-pins the polling threads when the caller has moved to another L3 domain (one
// sched_getcpu and a table look-up otherwise).  A domain without room for them: they go back to where the pool runs.
void pin_pollers_to_callers_l3(sv_handle *h, int hot) {
    const int cpu = sched_getcpu();
    if (cpu < 0) return;
    static std::mutex mu;
    static std::vector<int> l3_of;  // first CPU of the L3 domain of each CPU asked about so far (-3: not yet, -1: unknown)
    std::lock_guard<std::mutex> lk(mu);
    if ((int)l3_of.size() <= cpu) l3_of.resize(cpu + 1, -3);
    std::vector<int> cores;
    bool room = false, looked = false;
    if (l3_of[cpu] == -3) {
        int first = -1;
        room = l3_helper_cores(cpu, hot, &cores, &first);
        looked = true;
        l3_of[cpu] = first;
    }
    if (l3_of[cpu] < 0 || l3_of[cpu] == h->lat_pin_l3) return;
    if (!looked) {
        int first = -1;
        room = l3_helper_cores(cpu, hot, &cores, &first);
    }
    const int n = std::min(hot, (int)h->pool.size());
    bool near = room;
    for (int i = 0; i < n && near; i++) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cores[i], &one);
        near = pthread_setaffinity_np(h->pool[i].native_handle(), sizeof(cpu_set_t), &one) == 0;  // (a cpuset may refuse)
    }
    if (!near)
        for (int i = 0; i < n; i++) (void)pthread_setaffinity_np(h->pool[i].native_handle(), sizeof(cpu_set_t), &h->pool_cpus);
    // under the automatic policy the host stage is only shared with helpers that sit next to the caller (anywhere else it costs more than it saves)
    h->lat_near.store(near, std::memory_order_release);
    h->lat_pin_l3 = l3_of[cpu];
}

template <class T>
void dev_alloc(T *&p, size_t count) {
    HIP_TRY(hipMalloc((void **)&p, count * sizeof(T)));
}

void alloc_slot(sv_handle *h, Slot *sl) {
    const Dims &d = h->kp.d;
    const size_t cap = (size_t)h->chunk;
    SlotDev &s = sl->dev;
    s.cap = (int)cap;
    dev_alloc(s.grad, cap * grad_bytes_per_pair(h->kp));
    HIP_TRY(hipMemset(s.grad, 0, cap * grad_bytes_per_pair(h->kp)));  // the row margins are read (never used) and never written
    if (h->cfg.keep_debug && !h->dbg_desc) dev_alloc(h->dbg_desc, cap * 2 * d.N * 16);
    dev_alloc(s.dcan, cap * d.Wc * d.Hc);
    sl->blob_words = cap * (META_WORDS + (size_t)d.max_pts * 3 + 2 * (size_t)d.max_tri * 3 + 2 * ((size_t)d.max_pts + 1) + 64);
    dev_alloc(s.blob, sl->blob_words);
    dev_alloc(s.fsup, cap * (size_t)d.max_pts * 3);
    dev_alloc(s.fnsup, cap);
    if (h->dg_limit > h->dg_sub_max) {  // support lists beyond the LDS kernel (4K-sized lattices) go through a mesh in global memory
        size_t tb, xb, rb;
        delaunay_scratch_bytes(h->dg_limit, (int)cap * 2, &tb, &xb, &rb);
        uint8_t *t = nullptr;
        dev_alloc(t, tb);
        s.dg.tri = t;
        dev_alloc(s.dg.xy, xb / sizeof(int32_t));
        dev_alloc(s.dg.res, rb / sizeof(uint32_t));
        s.dg.cap = h->dg_limit;
        if (h->resident_ok) {  // resident chunks prepare their large sets on the device as well
            uint8_t *pr = nullptr;
            dev_alloc(pr, delaunay_prep_large_bytes(d.W, d.H, d.step, d.disp_max, h->dg_limit) * cap * 2);
            s.dg.prep = pr;
        }
    }
    {
        uint8_t *w = nullptr;
        dev_alloc(w, support_filter_ws_bytes(h->kp, (int)cap));
        s.flt_ws = w;
    }
    HIP_TRY(hipHostMalloc((void **)&sl->h_fsup, sizeof(int32_t) * cap * fsup_copy_pts(d) * 3, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&sl->h_fnsup, sizeof(int32_t) * cap, hipHostMallocDefault));
    dev_alloc(s.trirec, cap * 2 * d.max_tri);
    {
        uint8_t *r = nullptr;
        dev_alloc(r, cap * 2 * (size_t)d.max_tri * 36);
        s.rrec = r;
    }
    dev_alloc(s.tile_list, cap * 2 * raster_tiles(h->kp) * 512);
    dev_alloc(s.planes, cap * 2 * d.max_tri * 6);
    {  // cell masks + raster tile counters: one allocation, one clear per chunk
        uint8_t *w = nullptr;
        dev_alloc(w, grid_clear_bytes(h->kp, (int)cap));
        s.gmaskA = reinterpret_cast<uint32_t *>(w);
        s.tile_cnt = reinterpret_cast<int32_t *>(s.gmaskA + grid_masks_words(h->kp, (int)cap));
    }
    dev_alloc(s.gmaskB, cap * 2 * d.ncell * d.MW);
    dev_alloc(s.tri_id, cap * 2 * d.N);
    dev_alloc(s.wta, cap * 2 * d.N);
    dev_alloc(s.disp, cap * 2 * d.N);
    dev_alloc(s.tmp, cap * 2 * d.N);
    dev_alloc(s.csize, cap * 2 * d.N);
    {
        uint8_t *w = nullptr;
        const size_t bytes = ccl_ws_bytes(h->kp, (int)cap * 2);
        dev_alloc(w, bytes);
        HIP_TRY(hipMemset(w, 0, bytes));  // the overflow marks start cleared (no launch has epoch 0)
        s.ccl_ws = w;
    }
    HIP_TRY(hipHostMalloc((void **)&sl->h_dcan, sizeof(int16_t) * (cap * d.Wc * d.Hc + LATTICE_PAD), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&sl->h_blob, sizeof(int32_t) * sl->blob_words, hipHostMallocDefault));
    {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, sl->h_dcan, 0) == hipSuccess) sl->h_dcan_dev = static_cast<int16_t *>(dp);
        if (hipHostGetDevicePointer(&dp, sl->h_blob, 0) == hipSuccess) sl->h_blob_dev = static_cast<int32_t *>(dp);
        (void)hipGetLastError();
    }
    // throughput mode: waiting threads sleep on the event instead of spinning, so the cores go to the Delaunay pool
    const unsigned evf = hipEventDisableTiming | (h->block_sync ? hipEventBlockingSync : 0u);
    HIP_TRY(hipEventCreateWithFlags(&sl->ev_p1, evf));
    HIP_TRY(hipEventCreateWithFlags(&sl->ev_free, evf));
    HIP_TRY(hipEventCreateWithFlags(&sl->ev_sup, hipEventDisableTiming));
}

void free_slot(Slot *sl) {
    SlotDev &s = sl->dev;
    void *dptrs[] = {s.dg.tri, s.dg.xy, s.dg.res, s.dg.prep, s.grad, s.dcan, s.fsup, s.fnsup, s.flt_ws, s.blob, s.rrec, s.tile_list, s.trirec, s.planes, s.gmaskA, s.gmaskB, s.tri_id, s.wta, s.disp, s.tmp, s.csize, s.ccl_ws};
    for (void *p : dptrs)
        if (p) (void)hipFree(p);
    if (sl->h_dcan) (void)hipHostFree(sl->h_dcan);
    if (sl->h_blob) (void)hipHostFree(sl->h_blob);
    if (sl->h_fsup) (void)hipHostFree(sl->h_fsup);
    if (sl->h_fnsup) (void)hipHostFree(sl->h_fnsup);
    if (sl->ev_p1) (void)hipEventDestroy(sl->ev_p1);
    if (sl->ev_free) (void)hipEventDestroy(sl->ev_free);
    if (sl->ev_sup) (void)hipEventDestroy(sl->ev_sup);
    if (sl->d_in) (void)hipFree(sl->d_in);
    if (sl->d_out) (void)hipFree(sl->d_out);
    if (sl->h_in) (void)hipHostFree(sl->h_in);
    if (sl->h_out) (void)hipHostFree(sl->h_out);
    if (sl->d_out8) (void)hipFree(sl->d_out8);
    if (sl->h_out8) (void)hipHostFree(sl->h_out8);
    for (hipEvent_t e : {sl->ev_in, sl->ev_lr, sl->ev_p2, sl->ev_out})
        if (e) (void)hipEventDestroy(e);
}

void free_handle_resources(sv_handle *h) {
    (void)hipSetDevice(h->cfg.device);
    for (Slot *sl : h->slots) {
        free_slot(sl);
        delete sl;
    }
    h->slots.clear();
    for (TimingCtx *t : {&h->tc_issue, &h->tc_finish})
        for (hipEvent_t e : t->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    h->ev_pool.clear();
    if (h->sP1) (void)hipStreamDestroy(h->sP1);
    for (hipStream_t st : h->sPF)
        if (st) (void)hipStreamDestroy(st);
    for (hipStream_t st : h->sP2) (void)hipStreamDestroy(st);
    if (h->d_counters) (void)hipFree(h->d_counters);
    if (h->dbg_desc) (void)hipFree(h->dbg_desc);
    if (h->sIn) (void)hipStreamDestroy(h->sIn);
    if (h->sOut) (void)hipStreamDestroy(h->sOut);
    if (h->sOut2) (void)hipStreamDestroy(h->sOut2);
    delete h->dma;
    h->dma = nullptr;
}

// host-memory jobs: classify the caller's memory and make sure the staging buffers exist
int prepare_host_job(sv_handle *h, Job *job) {
    const Dims &d = h->kp.d;
    job->host = true;
    const size_t in_bytes = (size_t)job->batch * d.H * job->stride, out_bytes = (size_t)job->batch * d.Nm * sizeof(float);
    (void)hipSetDevice(h->cfg.device);
    job->pin_in = is_pinned_host(job->left, in_bytes) && is_pinned_host(job->right, in_bytes);
    job->pin_out = job->dmap ? is_pinned_host(job->dmap, (size_t)job->batch * d.Nm)
                             : is_pinned_host(job->d1, out_bytes) && (!job->d2 || is_pinned_host(job->d2, out_bytes));
    if (h->force_staging) job->pin_in = job->pin_out = false;  // tests (sv_debug_set "host_force_staging"): the pageable route with page-locked buffers
    try {
        ensure_host_staging(h, !job->pin_in, !job->pin_out, job->dmap != nullptr);
    } catch (const std::exception &e) {
        h->error = e.what();
        return SV_ERR_HIP;
    }
    return SV_OK;
}

int submit_job(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status, bool host = false, uint8_t *dmap = nullptr) {
    if (!h) return SV_ERR_ARG;
    if (!left || !right || (!d1 && !(host && dmap)) || batch < 0 || stride < h->cfg.width) {
        h->error = "bad argument (null pointer, negative batch, or stride < width)";
        return SV_ERR_ARG;
    }
    if (batch == 0) return SV_OK;
    Job *job = new Job();
    job->left = left;
    job->right = right;
    job->batch = batch;
    job->stride = stride;
    job->d1 = d1;
    job->d2 = d2;
    job->dmap = dmap;
    job->status = status;
    job->nchunks = (batch + h->chunk - 1) / h->chunk;
    if (host) {
        const int rc = prepare_host_job(h, job);
        if (rc != SV_OK) {
            delete job;
            return rc;
        }
    }
    if (!host) {  // one completion event per chunk (from the handle's pool; created on first use)
        (void)hipSetDevice(h->cfg.device);
        std::lock_guard<std::mutex> lk(h->mu);
        job->ev_done.assign((size_t)job->nchunks, nullptr);
        for (hipEvent_t &e : job->ev_done) {
            if (!h->ev_pool.empty()) {
                e = h->ev_pool.back();
                h->ev_pool.pop_back();
            } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming | (h->block_sync ? hipEventBlockingSync : 0u)) != hipSuccess) {
                e = nullptr;  // (sv_wait_batches then falls back to the streams for this batch)
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (h->live.empty()) h->failed = false;
        h->jobs.push_back(job);
        h->live.push_back(job);
        h->jobs_submitted++;
    }
    h->cv.notify_all();
    return SV_OK;
}

// Waits until every submitted batch is complete and its outputs are visible to every stream of the device.
int wait_jobs(sv_handle *h) {
    if (!h) return SV_ERR_ARG;
    {
        std::unique_lock<std::mutex> lk(h->mu);
        h->cv.wait(lk, [&] { return h->jobs_finished == h->jobs_submitted; });
    }
    (void)hipSetDevice(h->cfg.device);
    bool ok = hipStreamSynchronize(h->sP1) == hipSuccess;
    for (hipStream_t st : h->sPF)
        if (st) ok = (hipStreamSynchronize(st) == hipSuccess) && ok;
    for (hipStream_t st : h->sP2) ok = (hipStreamSynchronize(st) == hipSuccess) && ok;
    for (hipStream_t st : {h->sIn, h->sOut, h->sOut2})
        if (st) ok = (hipStreamSynchronize(st) == hipSuccess) && ok;
    if (!ok) note_error(h, "stream synchronisation failed");
    if (h->timing) {
        collect_timing(h, &h->tc_issue);
        collect_timing(h, &h->tc_finish);
    }
    std::lock_guard<std::mutex> lk(h->mu);
    for (Job *j : h->live) {
        for (hipEvent_t e : j->ev_done)
            if (e) h->ev_pool.push_back(e);
        delete j;
    }
    h->live.clear();
    return h->failed ? SV_ERR_HIP : SV_OK;
}

// Waits until the `n` oldest batches submitted since the last sv_wait are complete (their maps visible to every stream of the
// device / in the caller's host memory).  Later batches keep flowing through the pipeline meanwhile: a consumer - e.g. the
// chunked gather of bench.py - takes finished batches while the engine computes the next ones.  Batches complete in
// submission order.  Must not run concurrently with sv_wait (which releases the batch records).
int wait_first_jobs(sv_handle *h, int n) {
    if (!h) return SV_ERR_ARG;
    if (n <= 0) return SV_OK;
    std::vector<hipEvent_t> evs;
    bool need_streams = false;
    {
        std::unique_lock<std::mutex> lk(h->mu);
        if ((size_t)n > h->live.size()) {
            h->error = "sv_wait_batches: fewer batches have been submitted since the last sv_wait";
            return SV_ERR_ARG;
        }
        Job *j = h->live[(size_t)n - 1];
        h->cv.wait(lk, [&] { return (j->host ? j->drained : j->issued2) == j->nchunks; });
        // every chunk's second phase of the first n batches is enqueued and has its completion event recorded behind it: waiting on
        // those events - not on the streams - leaves the phase-2 work of LATER batches, which the finisher may have enqueued
        // already, out of the wait (the chunked gather of bench.py starts a chunk's collective as soon as that chunk is done)
        for (int q = 0; q < n; q++) {
            const Job *b = h->live[(size_t)q];
            if (b->host) continue;  // (host-memory batches are complete once drained)
            for (hipEvent_t e : b->ev_done) {
                if (e) evs.push_back(e);
                else need_streams = true;
            }
        }
    }
    (void)hipSetDevice(h->cfg.device);
    bool ok = true;
    for (hipEvent_t e : evs) ok = (wait_event(h, e) == hipSuccess) && ok;
    if (need_streams)
        for (hipStream_t st : h->sP2) ok = (hipStreamSynchronize(st) == hipSuccess) && ok;
    if (!ok) note_error(h, "waiting for the batches' completion events failed");
    return h->failed ? SV_ERR_HIP : SV_OK;
}

// Latency mode: one pair, chunk 1, nothing in flight.  The calling thread issues phase 1, spins on its event, runs the lattice
// filter and the left triangulation itself (the right one goes to a pool thread meanwhile), issues phase 2 and waits for it:
// no hand-over between the three control threads, which costs more than the kernels of a single pair.
int run_inline(sv_handle *h, const uint8_t *left, const uint8_t *right, int stride, float *d1, float *d2, int32_t *status, bool host) {
    Job job;
    job.left = left;
    job.right = right;
    job.batch = 1;
    job.stride = stride;
    job.d1 = d1;
    job.d2 = d2;
    job.status = status;
    job.nchunks = 1;
    const uint8_t *mirror_in = nullptr;  // pageable images: the device's view of the slot's page-locked mirror
    const auto t_prep = std::chrono::steady_clock::now();
    if (host) {
        const int rc = prepare_host_job(h, &job);
        if (rc != SV_OK) return rc;
        // One pair at a time has no use for staged copies: with page-locked memory the kernels address the caller's buffers themselves
        // (0.93 MB of gray rows in, 1.86 MB per map out: the link's time hides inside the kernels, and two copies with their
        // hand-overs go away).  Streamed batches do NOT do this - the matching kernels would sit on their LDS while the link
        // trickles (tools/zc_probe.py: 11 900 against 27 500 pairs/s).
        void *dp = nullptr;
        if (h->cfg.host_copies != 1 && job.pin_in && hipHostGetDevicePointer(&dp, const_cast<uint8_t *>(left), 0) == hipSuccess) {
            job.zc_left = static_cast<const uint8_t *>(dp);
            if (hipHostGetDevicePointer(&dp, const_cast<uint8_t *>(right), 0) == hipSuccess) job.zc_right = static_cast<const uint8_t *>(dp);
            else job.zc_left = nullptr;
        }
        if (h->cfg.host_copies != 1 && job.pin_out && hipHostGetDevicePointer(&dp, d1, 0) == hipSuccess) {
            job.zc_d1 = static_cast<float *>(dp);
            if (d2) {
                if (hipHostGetDevicePointer(&dp, d2, 0) == hipSuccess) job.zc_d2 = static_cast<float *>(dp);
                else job.zc_d1 = nullptr;
            }
        }
        // Pageable buffers (what the reference's cv::Mat hands over): the same without copies by the runtime - the images are packed into
        // the slot's page-locked mirror and read from there by k_sobel, the kernels write the maps into the page-locked mirror of the
        // outputs, and the pool's pollers help with the two memcpys.  (Staged DMA copies in front of and behind the kernels: +0.25 ms.)
        Slot *s0 = h->slots[0];
        if (h->cfg.host_copies != 1 && !job.pin_in && s0->h_in && hipHostGetDevicePointer(&dp, s0->h_in, 0) == hipSuccess) mirror_in = static_cast<const uint8_t *>(dp);
        if (h->cfg.host_copies != 1 && !job.pin_out && !job.dmap && s0->h_out && hipHostGetDevicePointer(&dp, s0->h_out, 0) == hipSuccess) {
            job.zc_d1 = static_cast<float *>(dp);
            if (d2) job.zc_d2 = job.zc_d1 + (size_t)s0->dev.cap * h->kp.d.Nm;
        }
        (void)hipGetLastError();
    }
    if (h->lat_pin) pin_pollers_to_callers_l3(h, latency_pollers(h));
    {  // the pace of the calls, for the pollers' sleep between frames (pool_main)
        const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const int64_t last = h->lat_last_start_ns;
        h->lat_last_start_ns = now;
        if (last) {
            const int64_t dt = now - last, p = h->lat_period_ns;
            h->lat_period_ns = (p > 0 && dt > p / 2 && dt < 2 * p) ? (3 * p + dt) / 4 : dt;
            h->lat_period_ns_pub.store(h->lat_period_ns, std::memory_order_relaxed);
            h->lat_next_expected_ns.store(now + h->lat_period_ns, std::memory_order_relaxed);
        }
    }
    Slot *s = h->slots[0];
    h->failed = false;
    s->job = &job;
    s->i0 = 0;
    s->n = 1;
    s->inline_done.store(0);
    s->grid_issued = false;
    s->inline_mode = true;
    (void)hipSetDevice(h->cfg.device);
    const LaunchHook saved = g_launch_hook;
    try {
        g_launch_hook.fn = h->timing ? timing_hook : nullptr;
        g_launch_hook.ctx = &h->tc_issue;
        using clk = std::chrono::steady_clock;
        clk::time_point tp[7];
        tp[0] = clk::now();
        if (host && mirror_in && !job.zc_left) {
            pack_images(h, s, std::max(2, h->pollers.load(std::memory_order_acquire)), COPY_PIECE / 8);
            s->in_left = mirror_in;
            s->in_right = mirror_in + (size_t)s->dev.cap * h->kp.d.N;
            s->in_pair = (size_t)h->kp.d.N;
            s->in_stride = h->kp.d.W;
        } else if (host && !job.zc_left) {
            upload_chunk(h, s, h->sP1, 2);  // on the phase-1 stream itself: in order, no event
        } else {
            set_chunk_inputs(h, s);
        }
        issue_phase1(h, s);
        tp[1] = clk::now();
        HIP_TRY(hipEventSynchronize(s->ev_p1));
        tp[2] = clk::now();
        s->blob_off.store((size_t)s->dev.cap * META_WORDS);
        s->pending.store(1);
        s->gpu_pct = 0;
        run_task(h, h->inline_scratch, Task{s, 0, -1});
        tp[3] = clk::now();
        while (!s->inline_done.load(std::memory_order_acquire)) __builtin_ia32_pause();
        tp[4] = clk::now();
        g_launch_hook.ctx = &h->tc_finish;
        issue_phase2(h, s, h->sP2[0]);
        tp[5] = clk::now();
        int delivered = 0;  // sides the caller's pageable maps have already
        if (host && !job.zc_d1) {
            download_chunk(h, s);
            deliver_maps(h, s, 3);
        } else if (host && h->nproc == 1 && job.d2) {  // zero-copy left map; the right map's DMA runs beside the post-processing kernels
            HIP_TRY(hipStreamWaitEvent(h->sOut, s->ev_lr, 0));
            download_maps(h, s, 1, h->sOut);
            if (!job.pin_out) {  // pageable: the right map goes from the mirror to the caller while the left one is still being post-processed
                HIP_TRY(hipStreamSynchronize(h->sOut));
                deliver_maps(h, s, std::max(3, h->pollers.load(std::memory_order_acquire)), 2);
                delivered = 2;
            }
            HIP_TRY(hipStreamSynchronize(h->sP2[0]));
            HIP_TRY(hipStreamSynchronize(h->sOut));
        } else {
            HIP_TRY(hipStreamSynchronize(h->sP2[0]));  // (maps written straight into page-locked host memory are visible now as well)
        }
        if (host && job.zc_d1 && !job.pin_out) deliver_maps(h, s, std::max(3, h->pollers.load(std::memory_order_acquire)), 3 & ~delivered);  // mirror -> the caller's pageable maps
        tp[6] = clk::now();
        if (h->lat_trace) {
            for (int i = 0; i < 6; i++) h->lat_ns[i] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(tp[i + 1] - tp[i]).count();
            h->lat_ns[6] += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(tp[0] - t_prep).count();  // classifying the caller's memory (host-memory calls)
            h->lat_calls++;
        }
    } catch (const std::exception &e) {
        note_error(h, e.what());
        while (s->pending.load() > 0 && !s->inline_done.load()) __builtin_ia32_pause();  // a queued right-side task still refers to the slot
    }
    g_launch_hook = saved;
    s->inline_mode = false;
    s->job = nullptr;
    if (h->timing) {
        collect_timing(h, &h->tc_issue);
        collect_timing(h, &h->tc_finish);
    }
    return h->failed ? SV_ERR_HIP : SV_OK;
}

int run_job(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status, bool host = false) {
    if (h && batch == 1 && h->chunk == 1 && !h->cfg.keep_debug && !h->gpu_filter && left && right && d1 && stride >= h->cfg.width && h->inline_ok) {
        bool idle;
        {
            std::lock_guard<std::mutex> lk(h->mu);
            idle = h->live.empty() && h->jobs.empty();
        }
        if (idle) return run_inline(h, left, right, stride, d1, d2, status, host);
    }
    const int rc = submit_job(h, left, right, batch, stride, d1, d2, status, host);
    if (rc != SV_OK) return rc;
    return wait_jobs(h);
}

}  // namespace

// Internal entry for the legacy path (legacy.cpp; not in the public header): like sv_process_batch_device for one pair, but the
// first GPU phase waits for `ready` on the device, so the caller does not have to synchronise the stream that produced the inputs.
int sv_internal_process_after(sv_handle *h, hipEvent_t ready, const uint8_t *left, const uint8_t *right, int stride, float *d1, float *d2) {
    if (!h) return SV_ERR_ARG;
    (void)hipSetDevice(h->cfg.device);
    if (ready && hipStreamWaitEvent(h->sP1, ready, 0) != hipSuccess) {
        h->error = "hipStreamWaitEvent failed";
        return SV_ERR_HIP;
    }
    return run_job(h, left, right, 1, stride, d1, d2, nullptr);
}

// Internal helper for the legacy path: two host-to-host copies (the frame's BGRA images into the page-locked staging buffer) shared with
// the handle's pool threads - the ones that poll next to the calling thread in latency mode.  3.7 MB on one thread is 120 - 150 us.
int sv_internal_copy2(sv_handle *h, void *dst_a, const void *src_a, void *dst_b, const void *src_b, size_t bytes_each) {
    if (!h || !dst_a || !src_a || !dst_b || !src_b) return SV_ERR_ARG;
    CopyList cl;
    cl.piece = COPY_PIECE / 2;
    cl.add(dst_a, src_a, bytes_each);
    cl.add(dst_b, src_b, bytes_each);
    cl.run(h, std::max(3, h->pollers.load(std::memory_order_acquire)));
    return SV_OK;
}

extern "C" {

void sv_params_init(sv_params *p, int setting) {
    if (!p) return;
    // elas.h:92-143
    p->disp_min = 0;
    p->disp_max = 255;
    p->support_texture = 10;
    p->candidate_stepsize = 5;
    p->incon_window_size = 5;
    p->incon_threshold = 5;
    p->incon_min_support = 5;
    p->grid_size = 20;
    p->beta = 0.02f;
    p->sigma = 1.0f;
    p->lr_threshold = 2;
    p->speckle_sim_threshold = 1.0f;
    p->speckle_size = 200;
    p->subsampling = 0;
    if (setting == SV_ROBOTICS) {
        p->support_threshold = 0.85f;
        p->add_corners = 0;
        p->gamma = 3.0f;
        p->sradius = 2.0f;
        p->match_texture = 1;
        p->ipol_gap_width = 3;
        p->filter_median = 0;
        p->filter_adaptive_mean = 1;
        p->postprocess_only_left = 1;
    } else {
        p->support_threshold = 0.95f;
        p->add_corners = 1;
        p->gamma = 5.0f;
        p->sradius = 3.0f;
        p->match_texture = 0;
        p->ipol_gap_width = 5000;
        p->filter_median = 1;
        p->filter_adaptive_mean = 0;
        p->postprocess_only_left = 0;
        if (setting == SV_DRIVER) {  // stereo_vision.cpp:307-311
            p->postprocess_only_left = 1;
            p->filter_adaptive_mean = 1;
        }
    }
}

int sv_create(const sv_params *params, const sv_config *cfg, sv_handle **out) {
    if (!params || !cfg || !out) {
        g_create_error = "null argument";
        return SV_ERR_ARG;
    }
    *out = nullptr;
    std::string err;
    int rc = validate(*params, *cfg, err);
    if (rc != SV_OK) {
        g_create_error = err;
        return rc;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (this library has no CPU fallback)";
        return SV_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        g_create_error = "device ordinal out of range";
        return SV_ERR_ARG;
    }
    sv_handle *h = new sv_handle();
    h->tc_issue.mask = h->tc_finish.mask = &h->timing_mask;
    h->p = *params;
    h->cfg = *cfg;
    apply_env_overrides(h->cfg);
    cfg = &h->cfg;  // from here on: the caller's configuration with the environment's overrides
    fill_kparams(h);
    int npool = cfg->n_workers > 0 ? cfg->n_workers : default_pool_size();
    // defaults: 64 pairs per launch, 8 slots, 5 phase-2 streams, scaled down so that the slots stay within a memory budget
    // (about 66 bytes per pixel per pair in flight: 31 MB at KITTI size, 0.55 GB at 4K)
    // (five: with the latency-chain kernels in 256-thread workgroups since round 5, 3 / 4 / 5 / 6 / 7 streams measure 47 300 / 48 000 / 48 750 / 48 700 / 48 200 pairs/s;
    //  four were the optimum until then - HISTORY.md)
    //  (4K, D = 192: 4 streams 2 900 - 2 930, 5 streams 2 760 - 2 790: large images keep four)
    int np2 = cfg->n_streams > 0 ? cfg->n_streams : ((size_t)cfg->width * cfg->height >= ((size_t)2 << 20) ? 4 : 5);
    int nslots = cfg->n_slots > 0 ? cfg->n_slots : 8;
    h->chunk = cfg->chunk > 0 ? cfg->chunk : 64;
    // Vertex sets the GPU triangulation takes: up to dg_sub_max whole in LDS; larger ones (4K lattices) through a mesh in the slots'
    // global-memory scratch, sized for the support lists a handle really sees (the bulk-copied head of a list, fsup_copy_pts: a
    // sixth of the lattice - a 4K pair has 21 000 of 330 000 lattice points) - not for the lattice.  A KITTI lattice's lists
    // (~2 000 points, 4 096 copied) fit the LDS kernel: no scratch there.  Sets beyond the limit are triangulated by the pool.
    h->dg_limit = h->dg_sub_max = delaunay_gpu_max_points();
    if (cfg->dg_sub_max > 0) h->dg_sub_max = std::max(6, std::min(h->dg_sub_max, cfg->dg_sub_max));  // experiments / tests
    if (!cfg->keep_debug) {
        const int want = std::min({h->kp.d.max_pts, delaunay_gpu_large_max_points(), 131072, fsup_copy_pts(h->kp.d)});
        if (want > h->dg_sub_max + h->dg_sub_max / 4 || (cfg->dg_sub_max > 0 && want > h->dg_sub_max)) {
            h->dg_limit = want;
            // Lists that have to be cut anyway (4K lattices: 21 000 - 30 000 vertices) are cut into subtrees of at most 1 024 vertices, not
            // 4 000: a subtree workgroup then holds 34 KB of LDS instead of 136 KB for a third of the time, and the kernels of the other
            // streams keep their occupancy - 2 460 against 2 210 pairs/s at 4K with two host threads (700: 2 480, 1 400: 2 390).
            if (cfg->dg_sub_max <= 0) h->dg_sub_max = 1024;
        }
    }
    if (cfg->dg_max_points > 0) h->dg_limit = std::min(h->dg_limit, std::max(cfg->dg_max_points, 16));  // tests: larger sets fall back to the pool
    // a set is cut into at most 2^cut_max subtrees of at most dg_sub_max vertices: with a lowered dg_sub_max (experiments) the cut path takes
    // less - a 4K list of 23 000 vertices with dg_sub_max = 350 overran the node-result table before this clamp existed
    h->dg_limit = std::max(h->dg_sub_max, (int)std::min<long>(h->dg_limit, (long)h->dg_sub_max << delaunay_gpu_cut_max()));
    if (cfg->chunk <= 0 || cfg->n_slots <= 0) {
        size_t free_b = 0, total_b = 0;
        (void)hipSetDevice(cfg->device);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)16 << 30;
        // (4K, one MI355X, round 2: chunk x slots 4 x 4 -> 1 500 pairs/s, 8 x 6 -> 2 020, 16 x 6 -> 2 440, 32 x 4 -> 2 550: the per-launch
        //  latency chains - lattice filter, speckle merges, GPU triangulation - want many pairs per launch more than many slots)
        const double budget = std::min(64.0 * (1 << 30), 0.25 * (double)free_b);
        double per_pair = 66.0 * (double)h->kp.d.N + 4.0e6;
        if (h->dg_limit > h->dg_sub_max)  // two vertex sets per pair: the cut path's mesh and (resident chunks) the preparation's arrays
            per_pair += (double)delaunay_scratch_bytes(h->dg_limit, 2, nullptr, nullptr, nullptr) + 2.0 * (double)delaunay_prep_large_bytes(h->kp.d.W, h->kp.d.H, h->kp.d.step, h->kp.d.disp_max, h->dg_limit);
        while ((double)h->chunk * nslots * per_pair > budget) {
            if (cfg->chunk <= 0 && h->chunk > 16)
                h->chunk = (h->chunk + 1) / 2;
            else if (cfg->n_slots <= 0 && nslots > 4)
                nslots--;
            else if (cfg->chunk <= 0 && h->chunk > 1)
                h->chunk = (h->chunk + 1) / 2;
            else if (cfg->n_slots <= 0 && nslots > 2)
                nslots--;
            else
                break;
        }
    }
    if (cfg->keep_debug) {
        nslots = 1;
        np2 = 1;
        h->chunk = 1;
    }
    if (h->chunk < 4 && cfg->gpu_lattice_filter != 1) h->gpu_filter = false;
    // Who triangulates (cfg.gpu_triangulation): the pool, the GPU kernels (delaunay_gpu.hip), or both.  Automatic: all on the GPU with one
    // or two pool threads, else balanced - the dispatcher moves a share of the chunks to the GPU while the pool falls behind it.  Never
    // on the GPU with keep_debug (the parity tests read the host's triangle lists) and not for single pairs (faster on the host).
    // Sustained on one MI355X, round 4 (resident GPU share): GPU alone 43 000 pairs/s with ONE host thread (20 400 in round 3, whose
    // vertex orders came from the pool), balanced 45 400 with 14 threads.
    const bool gpu_capable = !cfg->keep_debug && h->chunk >= 4;
    const int mode = cfg->gpu_triangulation;
    h->gpu_delaunay = gpu_capable && (mode == 1 || (mode == 0 && npool < 3));
    h->gpu_delaunay_pct = h->gpu_delaunay ? 100 : 0;
    if (mode == 3) h->gpu_delaunay_pct = gpu_capable ? std::max(0, std::min(100, cfg->gpu_triangulation_pct)) : 0;
    if (h->gpu_delaunay_pct >= 100) h->gpu_delaunay = true;
    // no fixed share: it follows the pool's backlog (42 600 against 40 100 pairs/s with 14 threads in round 2)
    h->gpu_share_auto = !h->gpu_delaunay && (mode == 0 || mode == 4) && gpu_capable;
    h->share_sliced = cfg->share_sliced != 0;
    h->latency_split = cfg->latency_split == 3 ? 0 : std::max(0, std::min(cfg->latency_split, 2));  // (0 = automatic: resolved below, once the pool's size is known)
    // start where the balance was measured to settle (4 ... 14 threads); a handle with fewer than four slots cannot build up a backlog
    if (h->gpu_share_auto && nslots >= 4) h->auto_pct = std::max(0, std::min(95, 117 - 7 * npool));
    if (!(h->gpu_delaunay || h->gpu_share_auto || h->gpu_delaunay_pct > 0)) h->dg_limit = h->dg_sub_max;  // the pool triangulates everything: no scratch
    // The GPU's share of the chunks is "resident" - support lists never leave the device, preparation and triangulation in one kernel of
    // phase 1 - where the lattice filter runs on the GPU and a pair's lists fit the LDS kernel (KITTI-sized lattices: ~2 000 points; a
    // 4K lattice's 21 000 take the cut path: k_dg_prepare_large_blob orders them in the slot's scratch).  cfg.resident = 2: the round-3 path.
    // Lists beyond the LDS kernel (dg_limit > dg_sub_max: the slots have the cut path's scratch) are prepared in global memory.
    h->resident_ok = cfg->resident != 2 && h->gpu_filter && gpu_capable && (h->gpu_delaunay || h->gpu_share_auto || h->gpu_delaunay_pct > 0) &&
                     std::min(h->kp.d.max_pts, fsup_copy_pts(h->kp.d)) <= std::max(h->dg_sub_max + h->dg_sub_max / 4, h->dg_limit) && h->kp.d.disp_max + h->kp.d.W < 30000 &&
                     h->dg_limit < 65536;
    {  // the LDS path's limit at this image size: the bit maps of the preparation grow with the image (a 4K lattice's do not fit at all)
        // (156 KB: the kernel has ~1 KB of static LDS beside the dynamic request, 160 KB in all)
        const Dims &d = h->kp.d;
        int lo = 0, hi = std::min(h->dg_sub_max, delaunay_prep_max_points());
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            if (mid >= 3 && delaunay_resident_lds_bytes(d.W, d.H, d.step, d.disp_max, mid) <= 156 * 1024) lo = mid; else hi = mid - 1;
        }
        h->resident_lds_max = lo < 3 ? 0 : lo;
    }
    h->ns_bound.store(std::min(h->dg_sub_max, delaunay_prep_max_points()));
    h->shared_pct.store(h->auto_pct);
    // Throughput mode (chunk >= 4): the waiting threads ask the event and nap.  hipEventSynchronize on a hipEventBlockingSync event was
    // measured at 0.9 of a core in the dispatcher plus 0.8 in the issuer while it waits (ROCm 7.2; tools/thread_cpu.py: 2.8 cores busy
    // against 1.1 at the same rate with one pool thread) - two of the two CPUs a rank has on a node shared by eight.
    h->block_sync = cfg->event_sync == 1 || cfg->event_sync == 3 || (cfg->event_sync == 0 && h->chunk >= 4);
    h->poll_sync = cfg->event_sync == 3 || (cfg->event_sync == 0 && h->chunk >= 4);
    try {
        HIP_TRY(hipSetDevice(cfg->device));
        HIP_TRY(hipStreamCreateWithFlags(&h->sP1, hipStreamNonBlocking));
        h->n_pf = (h->resident_ok && h->dg_limit > h->dg_sub_max) ? 4 : 2;
        {
            int v = 0;
            if (env_int("SV_PF_STREAMS", &v)) h->n_pf = std::max(1, std::min(v, 8));  // experiments
        }
        for (int i = 0; i < h->n_pf; i++) HIP_TRY(hipStreamCreateWithFlags(&h->sPF[i], hipStreamNonBlocking));
        for (int i = 0; i < np2; i++) {
            hipStream_t st;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            h->sP2.push_back(st);
        }
        for (int i = 0; i < nslots; i++) {
            Slot *sl = new Slot();
            sl->id = i;
            h->slots.push_back(sl);
            alloc_slot(h, sl);
        }
    } catch (const std::exception &e) {
        g_create_error = e.what();
        free_handle_resources(h);
        delete h;
        return SV_ERR_HIP;
    }
    // Threads on the CPUs of the GPU's NUMA node - only when that node has room for them within this process's mask: a cpuset that spans
    // sockets with few CPUs on the GPU's node (say 16 CPUs, 2 of them there) would otherwise squeeze ~20 threads onto those few.
    cpu_set_t node_cpus;
    const cpu_set_t *bind = nullptr;
    if (cfg->affinity != 2 && gpu_node_cpus(cfg->device, &node_cpus)) {
        bool have_quota = false;
        const int share = std::max(1, (int)host_cpu_share(&have_quota));
        if (CPU_COUNT(&node_cpus) >= std::min(npool + 2, share)) bind = &node_cpus;
    }
    h->node_bound = bind != nullptr;
    if (bind) {
        h->pool_cpus = node_cpus;
    } else {
        CPU_ZERO(&h->pool_cpus);
        if (sched_getaffinity(0, sizeof(cpu_set_t), &h->pool_cpus) != 0)
            for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &h->pool_cpus);
    }
    if (h->chunk == 1) {
        // single pairs: the two triangulations in halves when the helpers can sit next to the calling thread (see l3_helper_cores)
        std::vector<int> cores;
        int first = -1;
        const int policy = cfg->latency_split;
        h->lat_auto = policy == 0;
        h->lat_pin = cfg->affinity != 2 && policy != 3;
        if (policy == 0) {  // quarters need seven helpers beside the caller's core, halves four
            h->latency_split = 0;
            if (h->lat_pin && npool >= 4 && l3_helper_cores(sched_getcpu(), 4, &cores, &first)) h->latency_split = (npool >= 7 && cores.size() >= 7) ? 2 : 1;
            if (!h->latency_split) h->lat_pin = 0;
        } else {
            h->latency_split = policy == 3 ? 0 : policy;
        }
    }
    for (int i = 0; i < npool; i++) {
        HostScratch *sc = new HostScratch();
        h->scratch.push_back(sc);
        h->pool.emplace_back(pool_main, h, sc, i);
        bind_thread(h->pool.back(), bind);
        name_thread(h->pool.back(), "sv-pool");
    }
    h->inline_scratch = new HostScratch();
    h->inline_ok = cfg->inline_latency_path != 2;
    {
        int v = 0;
        h->lat_trace = env_int("SV_LAT_TRACE", &v);  // (also sv_debug_set "lat_trace")
        if (env_int("SV_DMA_ENGINES", &v)) h->dma_engines_override = (uint32_t)v;
        if (const char *path = getenv("SV_CHUNK_TRACE")) {
            h->chunk_trace_on = *path != 0;
            h->chunk_trace_path = path;
        }
        h->pool_sleep = env_int("SV_POOL_SLEEP", &v);
        h->lat_runtime_copies = env_int("SV_LAT_RUNTIME_COPIES", &v) && v;
        h->lat_filter_alone = env_int("SV_LAT_FILTER_ALONE", &v) && v;
        if (env_int("SV_LAT_WAKE_LEAD_US", &v) && v > 0) h->lat_wake_lead_ns = (int64_t)v * 1000;
        if (env_int("SV_LATENCY_PIN", &v)) h->lat_pin = v;  // (experiments: shared triangulations without / with the helpers next to the caller)
        if (env_int("SV_DG_MARGIN", &v)) h->ns_margin_pct = std::max(0, v);
    }
    h->t_issue = std::thread(issuer_main, h);
    h->t_dispatch = std::thread(dispatcher_main, h);
    h->t_finish = std::thread(finisher_main, h);
    h->t_drain = std::thread(drainer_main, h);
    h->t_deliver = std::thread(deliverer_main, h);
    for (std::thread *t : {&h->t_issue, &h->t_dispatch, &h->t_finish, &h->t_drain, &h->t_deliver}) bind_thread(*t, bind);
    name_thread(h->t_issue, "sv-issue"), name_thread(h->t_dispatch, "sv-dispatch"), name_thread(h->t_finish, "sv-finish"), name_thread(h->t_drain, "sv-drain"), name_thread(h->t_deliver, "sv-deliver");
    *out = h;
    return SV_OK;
}

int sv_destroy(sv_handle *h) {
    if (!h) return SV_ERR_ARG;
    (void)wait_jobs(h);
    if (h->chunk_trace_on && !h->chunk_trace.empty()) {
        if (FILE *f = fopen(h->chunk_trace_path.c_str(), "a")) {
            fprintf(f, "# handle %p: slot stage ns\n", (void *)h);
            for (const sv_handle::TraceRec &r : h->chunk_trace) fprintf(f, "%d %s %lld\n", r.slot, kTraceStageNames[r.stage], (long long)r.ns);
            fclose(f);
        }
    }
    if (h->lat_trace && h->issue_chunks > 0)
        fprintf(stderr, "issuer, %ld chunks, ms of wall clock per chunk: sobel + support launches %.3f, filter launches %.3f, triangulation launch %.3f, copies %.3f, event record %.3f, waiting for a slot %.3f\n",
                h->issue_chunks, 1e-6 * h->issue_ns[0] / (double)h->issue_chunks, 1e-6 * h->issue_ns[1] / (double)h->issue_chunks, 1e-6 * h->issue_ns[2] / (double)h->issue_chunks,
                1e-6 * h->issue_ns[3] / (double)h->issue_chunks, 1e-6 * h->issue_ns[4] / (double)h->issue_chunks, 1e-6 * h->issue_ns[5] / (double)h->issue_chunks);
    if (h->lat_trace && h->drain_chunks > 0)
        fprintf(stderr, "host-memory path, %ld chunks, ms per chunk: wait for phase 2 %.3f, download %.3f, deliver %.3f\n", h->drain_chunks,
                1e-6 * h->drain_ns[0] / (double)h->drain_chunks, 1e-6 * h->drain_ns[1] / (double)h->drain_chunks, 1e-6 * h->drain_ns[2] / (double)h->drain_chunks);
    if (h->lat_trace && h->lat_calls > 0) {
        static const char *names[6] = {"enqueue phase 1", "wait phase 1", "filter + left triangulation", "wait right triangulation", "enqueue phase 2", "wait phase 2 (+ downloads)"};
        fprintf(stderr, "latency path, %ld calls, us per call:", h->lat_calls);
        for (int i = 0; i < 6; i++) fprintf(stderr, "  %s %.1f", names[i], 1e-3 * h->lat_ns[i] / (double)h->lat_calls);
        fprintf(stderr, "  before them, classifying the caller's memory %.1f", 1e-3 * h->lat_ns[6] / (double)h->lat_calls);
        fprintf(stderr, "  (of the third: lattice filters %.1f, hand-over %.1f, left triangulation %.1f)", 1e-3 * h->lat_sub_ns[0] / (double)h->lat_calls, 1e-3 * h->lat_sub_ns[1] / (double)h->lat_calls,
                1e-3 * h->lat_sub_ns[2] / (double)h->lat_calls);
        fprintf(stderr, "\n");
    }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->quit = true;
    }
    h->cv.notify_all();
    for (std::thread *t : {&h->t_issue, &h->t_dispatch, &h->t_finish, &h->t_drain, &h->t_deliver})
        if (t->joinable()) t->join();
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        h->pool_quit = true;
    }
    h->qcv.notify_all();
    for (std::thread &t : h->pool)
        if (t.joinable()) t.join();
    for (HostScratch *sc : h->scratch) delete sc;
    delete h->inline_scratch;
    free_handle_resources(h);
    delete h;
    return SV_OK;
}

const char *sv_last_error(const sv_handle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int sv_process_batch_device(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    return run_job(h, left, right, batch, stride, d1, d2, status);
}

int sv_submit_batch_device(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    return submit_job(h, left, right, batch, stride, d1, d2, status);
}

int sv_wait(sv_handle *h) { return wait_jobs(h); }

int sv_wait_batches(sv_handle *h, int n) { return wait_first_jobs(h, n); }

int sv_query(const sv_handle *h, int what) {
    if (!h) return SV_ERR_ARG;
    switch (what) {
        case SV_Q_HOST_THREADS: return (int)h->pool.size();
        case SV_Q_CHUNK: return h->chunk;
        case SV_Q_SLOTS: return (int)h->slots.size();
        case SV_Q_GPU_LATTICE_FILTER: return h->gpu_filter ? 1 : 0;
        case SV_Q_GPU_TRIANGULATION: return h->gpu_delaunay ? 1 : 0;
        case SV_Q_NUMA_BOUND: return h->node_bound ? 1 : 0;
        case SV_Q_RESIDENT: return h->resident_ok ? 1 : 0;
        case SV_Q_HOST_COPIES: return h->host_copies_mode;
        case SV_Q_LATENCY_SPLIT:  // (automatic policy: what the last single-pair call could do - no sharing while the helpers cannot sit next to the caller)
            return h->chunk != 1 ? 0 : (h->lat_auto && h->lat_pin_l3 != -2 && !h->lat_near.load()) ? 0 : h->latency_split;
        case SV_Q_GPU_TRIANGULATION_FALLBACKS: return (int)std::min<int64_t>(h->gpu_tri_fallbacks.load(), 0x7FFFFFFF);
        case SV_Q_GPU_TRIANGULATION_SHARE: {
            const int64_t all = h->tri_pairs.load(), g = h->gpu_tri_pairs.load();
            return all > 0 ? (int)((g * 1000 + all / 2) / all) : 0;
        }
        default: return SV_ERR_ARG;
    }
}

// Host memory in and out: the batch streams through the same pipeline, chunk by chunk - images up on a copy stream while
// earlier chunks compute, maps down on another while later chunks compute (a pair = gray L+R in host memory ->
// D1 back in host memory).  No allocation, no device-wide synchronisation per call (staging is allocated at the first call).
int sv_process_batch_host(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    return run_job(h, left, right, batch, stride, d1, d2, status, true);
}

int sv_submit_batch_host(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    return submit_job(h, left, right, batch, stride, d1, d2, status, true);
}

int sv_submit_batch_host_dmap(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, uint8_t *dmap, int32_t *status) {
    if (h && !dmap) {
        h->error = "bad argument (null dmap)";
        return SV_ERR_ARG;
    }
    return submit_job(h, left, right, batch, stride, nullptr, nullptr, status, true, dmap);
}

int sv_process_batch_host_dmap(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, uint8_t *dmap, int32_t *status) {
    const int rc = sv_submit_batch_host_dmap(h, left, right, batch, stride, dmap, status);
    return rc != SV_OK ? rc : wait_jobs(h);
}

void *sv_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void sv_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int sv_elas_process(sv_handle *h, const uint8_t *I1, const uint8_t *I2, float *D1, float *D2, const int32_t *dims) {
    if (!h || !dims) return SV_ERR_ARG;
    if (dims[0] != h->cfg.width || dims[1] != h->cfg.height) {
        h->error = "dims do not match the handle's width/height";
        return SV_ERR_ARG;
    }
    return sv_process_batch_host(h, I1, I2, 1, dims[2], D1, D2, nullptr);
}

int sv_debug_set(sv_handle *h, const char *key, int value) {
    if (!h || !key) return SV_ERR_ARG;
    (void)wait_jobs(h);  // the kernels' parameter blocks only change while nothing is in flight
    const std::string k(key);
    const bool gpu_filter = h->gpu_filter;
    if (k == "ccl_cap") {
        h->dbg_ccl_cap = std::max(0, value);
        fill_kparams(h);
        h->gpu_filter = gpu_filter;
    } else if (k == "rt_cap") {
        h->dbg_rt_cap = value;
        fill_kparams(h);
        h->gpu_filter = gpu_filter;
    } else if (k == "host_force_staging") {
        h->force_staging = value != 0;
    } else if (k == "dma_selftest_fail") {  // (before the first host-memory batch) the lanes' self-test is taken as failed: the fallback to the runtime's copies
        h->dbg_dma_fail = value != 0;
    } else if (k == "ns_bound") {
        h->ns_bound.store(std::max(3, std::min(value, h->dg_sub_max)));
    } else if (k == "pool_sleep") {
        h->pool_sleep = value != 0;
    } else if (k == "lat_filter_alone") {
        h->lat_filter_alone = value != 0;
    } else if (k == "lat_runtime_copies") {
        h->lat_runtime_copies = value != 0;
    } else if (k == "latency_pin") {
        h->lat_pin = value != 0;
    } else if (k == "lat_trace") {
        h->lat_trace = value != 0;
    } else {
        h->error = "sv_debug_set: unknown key";
        return SV_ERR_ARG;
    }
    return SV_OK;
}

int sv_debug_counters(sv_handle *h, int mode, uint64_t *out) {
    if (!h) return SV_ERR_ARG;
    (void)wait_jobs(h);  // the slots' counter pointer only changes while nothing is in flight
    if (hipSetDevice(h->cfg.device) != hipSuccess) return SV_ERR_HIP;
    if (mode == 1) {
        if (!h->d_counters && hipMalloc((void **)&h->d_counters, sizeof(unsigned long long) * CNT_COUNT) != hipSuccess) return SV_ERR_HIP;
        if (hipMemset(h->d_counters, 0, sizeof(unsigned long long) * CNT_COUNT) != hipSuccess) return SV_ERR_HIP;
    }
    if (mode == 0 || mode == 1)
        for (Slot *sl : h->slots) sl->dev.counters = mode ? h->d_counters : nullptr;
    if (out) {
        unsigned long long v[CNT_COUNT] = {0};
        if (h->d_counters && hipMemcpy(v, h->d_counters, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return SV_ERR_HIP;
        out[0] = v[CNT_DENSE_CANDIDATES];
        out[1] = v[CNT_DENSE_PIXELS];
        out[2] = v[CNT_SUPPORT_ENERGIES];
        out[3] = v[CNT_DENSE_BAND_FULL];
        out[4] = v[CNT_DENSE_BAND_PART];
        out[5] = v[CNT_DENSE_BAND_SLOW];
        out[6] = v[CNT_DENSE_GRID_WAVE_TRIPS];
        out[7] = v[CNT_DENSE_GRID_LANE_TRIPS];
    }
    return SV_OK;
}

long sv_debug_size(sv_handle *h, const char *name) {
    if (!h || !name) return -1;
    auto it = h->dbg.find(name);
    return it == h->dbg.end() ? -1 : (long)it->second.size();
}

long sv_debug_get(sv_handle *h, const char *name, void *out, long cap) {
    if (!h || !name) return -1;
    auto it = h->dbg.find(name);
    if (it == h->dbg.end()) return -1;
    const long n = (long)it->second.size();
    if (n > cap) return -2;
    memcpy(out, it->second.data(), n);
    return n;
}

int sv_kernel_times(sv_handle *h, const char **names, double *total_ms, int64_t *calls, int cap) {
    if (!h) return 0;
    std::lock_guard<std::mutex> g(h->tmu);
    for (int i = 0; i < K_COUNT && i < cap; i++) {
        if (names) names[i] = kernel_name(i);
        if (total_ms) total_ms[i] = h->k_ms[i];
        if (calls) calls[i] = h->k_calls[i];
    }
    // two pseudo entries: CPU time of the host stage (summed over pool threads), calls = pairs
    static const char *host_names[2] = {"host:lattice_filter", "host:delaunay_x2"};
    const int64_t ns[2] = {h->host_filter_ns.load(), h->host_delaunay_ns.load()};
    for (int j = 0; j < 2 && K_COUNT + j < cap; j++) {
        if (names) names[K_COUNT + j] = host_names[j];
        if (total_ms) total_ms[K_COUNT + j] = 1e-6 * (double)ns[j];
        if (calls) calls[K_COUNT + j] = h->host_tasks.load();
    }
    return K_COUNT + 2;
}

void sv_kernel_times_reset(sv_handle *h) {
    if (!h) return;
    std::lock_guard<std::mutex> g(h->tmu);
    for (int i = 0; i < K_COUNT; i++) {
        h->k_ms[i] = 0;
        h->k_calls[i] = 0;
    }
    h->host_filter_ns = 0;
    h->host_delaunay_ns = 0;
    h->host_tasks = 0;
}

void sv_kernel_timing_enable(sv_handle *h, int on) {
    if (h) h->timing = on != 0;
}

int sv_kernel_timing_select(sv_handle *h, const char *names) {
    if (!h) return SV_ERR_ARG;
    if (!names || !*names) {
        h->timing_mask.store(0xFFFFFFFFu);
        return SV_OK;
    }
    uint32_t mask = 0;
    std::string all(names);
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t end = std::min(all.find(',', pos), all.size());
        const std::string one = all.substr(pos, end - pos);
        bool found = one.empty();
        for (int i = 0; i < K_COUNT; i++)
            if (one == kernel_name(i)) {
                mask |= 1u << i;
                found = true;
            }
        if (!found) return SV_ERR_ARG;
        pos = end + 1;
    }
    h->timing_mask.store(mask);
    return SV_OK;
}

// Test hook (no GPU needed): the host pool a handle would get by default in this process - CPU quota / affinity mask / LOCAL_WORLD_SIZE;
// ignore_quota != 0 takes the branch of a host without a cgroup quota whatever this one has.
int sv_default_host_threads(int ignore_quota) { return default_pool_size(ignore_quota != 0); }

int sv_host_support_filter(const sv_params *p, int16_t *dcan, int width, int height, int32_t *support, int cap) {
    if (!p || !dcan || !support) return SV_ERR_ARG;
    return support_filter(*p, dcan, width, height, support, cap);
}

int sv_host_support_filter_threads(const sv_params *p, int16_t *dcan, int width, int height, int32_t *support, int cap, int threads) {
    if (!p || !dcan || !support || threads < 1) return SV_ERR_ARG;
    return support_filter_threads(*p, dcan, width, height, support, cap, threads);
}

int sv_host_delaunay(const int32_t *xy, int n, int32_t *tri_out, int cap) {
    if (!xy || !tri_out) return SV_ERR_ARG;
    Delaunay dl;
    return dl.triangulate(xy, n, tri_out, cap);
}

// Test hook: the same triangulation with the divide-and-conquer phase on the GPU (delaunay_gpu.hip); sort, duplicate scan and
// k-d ordering stay on the host.  `reps` identical sets are triangulated in one launch (throughput measurements); the first
// result is returned.  Returns the triangle count, or < 0.
int sv_gpu_delaunay(const int32_t *xy, int n, int32_t *tri_out, int cap, int reps, double *kernel_ms) {
    if (!xy || !tri_out || n < 3 || reps < 1) return SV_ERR_ARG;
    // sets of more than sub_max points take the cut path (subtrees in LDS, upper merges in a global-memory mesh);
    // SV_DG_SUBMAX lowers the limit so that tests reach deep cuts with small sets
    int sub_max = delaunay_gpu_max_points(), v = 0;
    if (env_int("SV_DG_SUBMAX", &v)) sub_max = std::max(6, std::min(sub_max, v));
    const bool large = n > sub_max;
    if (large && (n > delaunay_gpu_large_max_points() || (size_t)n * reps > ((size_t)1 << 24))) return SV_ERR_UNSUPPORTED;
    Delaunay dl;
    std::vector<int32_t> ids(n);
    const int m = dl.kd_ordered_ids(xy, n, ids.data());
    if (m < 0) return m;
    if (m < 3) return 0;
    int32_t *d_order = nullptr, *d_xy = nullptr, *d_tri = nullptr, *d_cnt = nullptr;
    int4 *d_sets = nullptr;
    DelaunayScratch scr;
    int rc = SV_OK, nt = 0;
    int narrow = 1;  // all coordinate differences below 2^14: the kernels' 32-bit in-circle terms are exact
    {
        int32_t lo[2] = {xy[0], xy[1]}, hi[2] = {xy[0], xy[1]};
        for (int i = 0; i < n; i++)
            for (int c = 0; c < 2; c++) lo[c] = std::min(lo[c], xy[2 * i + c]), hi[c] = std::max(hi[c], xy[2 * i + c]);
        if ((int64_t)hi[0] - lo[0] >= (1 << 14) || (int64_t)hi[1] - lo[1] >= (1 << 14)) narrow = 0;
        if (lo[0] < -32768 || lo[1] < -32768 || hi[0] > 32767 || hi[1] > 32767) return SV_ERR_UNSUPPORTED;  // 16-bit coordinates in the LDS mesh
    }
    const size_t tri_words = (size_t)3 * 2 * n;
    try {
        HIP_TRY(hipMalloc((void **)&d_order, sizeof(int32_t) * n));
        HIP_TRY(hipMalloc((void **)&d_xy, sizeof(int32_t) * 2 * n));
        HIP_TRY(hipMalloc((void **)&d_tri, sizeof(int32_t) * tri_words * reps));
        HIP_TRY(hipMalloc((void **)&d_cnt, sizeof(int32_t) * reps));
        HIP_TRY(hipMalloc((void **)&d_sets, sizeof(int4) * reps));
        std::vector<int4> sets(reps);
        for (int r = 0; r < reps; r++) sets[r] = make_int4(0, m, n, (int)(tri_words * r));
        HIP_TRY(hipMemcpy(d_order, ids.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_xy, xy, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_sets, sets.data(), sizeof(int4) * reps, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, nullptr));
        if (large) {
            size_t tb, xb, rb;
            delaunay_scratch_bytes(n, reps, &tb, &xb, &rb);
            HIP_TRY(hipMalloc(&scr.tri, tb));
            HIP_TRY(hipMalloc((void **)&scr.xy, xb));
            HIP_TRY(hipMalloc((void **)&scr.res, rb));
            scr.cap = n;
            HIP_TRY(hipEventRecord(e0, nullptr));  // (after the allocations)
            if (launch_delaunay_gpu_large(d_sets, reps, d_order, d_xy, d_tri, d_cnt, m, sub_max, scr, narrow, nullptr) != 0) throw std::runtime_error("k_dgl launch failed");
        } else {
            long long *d_clk = nullptr;
            const bool want_clk = env_int("SV_DG_LEVEL_CLOCK", nullptr);  // tools/gpu_delaunay_check.py: time per tree depth (100 MHz ticks) to stderr
            if (want_clk) HIP_TRY(hipMalloc((void **)&d_clk, sizeof(long long) * 32));
            if (want_clk) HIP_TRY(hipMemset(d_clk, 0, sizeof(long long) * 32));
            if (launch_delaunay_gpu(d_sets, reps, d_order, d_xy, d_tri, d_cnt, delaunay_gpu_lds_bytes(m, n), narrow, nullptr, d_clk) != 0) throw std::runtime_error("k_delaunay launch failed");
            if (want_clk) {
                long long c[32];
                HIP_TRY(hipMemcpy(c, d_clk, sizeof(c), hipMemcpyDeviceToHost));
                (void)hipFree(d_clk);
                fprintf(stderr, "triangulation of %d vertices, us per tree depth (deepest first):", m);
                long long prev = c[0];
                for (int d = 30; d >= 0; d--)
                    if (c[1 + d]) {
                        fprintf(stderr, " d%d %.1f", d, 0.01 * (double)(c[1 + d] - prev));
                        prev = c[1 + d];
                    }
                fprintf(stderr, "\n");
            }
        }
        HIP_TRY(hipEventRecord(e1, nullptr));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        if (kernel_ms) *kernel_ms = ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        HIP_TRY(hipMemcpy(&nt, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (nt > cap) {
            rc = -1;
        } else {
            HIP_TRY(hipMemcpy(tri_out, d_tri, sizeof(int32_t) * 3 * (size_t)nt, hipMemcpyDeviceToHost));
        }
    } catch (const std::exception &e) {
        g_create_error = e.what();
        rc = SV_ERR_HIP;
    }
    for (void *p : {(void *)d_order, (void *)d_xy, (void *)d_tri, (void *)d_cnt, (void *)d_sets, scr.tri, (void *)scr.xy, (void *)scr.res})
        if (p) (void)hipFree(p);
    return rc == SV_OK ? nt : rc;
}

// Test hooks: the preparation of a vertex set (sort, duplicate scan, k-d order) on the host (what Delaunay::prepare leaves: the ids of
// the surviving vertices in the order the recursion consumes them) and on the GPU (delaunay_gpu.hip: dg_prepare, for vertices on the
// support lattice of a width x height image).  Both return m and write m ids; the GPU form returns -1 for a set it leaves to the host
// (coincident points, or vertices outside the lattice's bit maps).
int sv_host_kd_order(const int32_t *xy, int n, int32_t *ids_out) {
    if (!xy || !ids_out || n < 0) return SV_ERR_ARG;
    Delaunay dl;
    return dl.kd_ordered_ids(xy, n, ids_out);
}

int sv_gpu_kd_order(const int32_t *xy, const int32_t *disp, int n, int width, int height, int step, int disp_max, int32_t *ids_out) {
    if (!xy || !ids_out || n < 3 || n > delaunay_prep_max_points() || step < 1 || width < 1 || height < 1 || disp_max < 0) return SV_ERR_ARG;
    int32_t *d_xy = nullptr, *d_ord = nullptr, *d_dsp = nullptr;
    int rc = SV_ERR_HIP;
    std::vector<int32_t> ord((size_t)n + 1, 0);
    if (hipMalloc((void **)&d_xy, sizeof(int32_t) * 2 * n) == hipSuccess && hipMalloc((void **)&d_ord, sizeof(int32_t) * ((size_t)n + 1)) == hipSuccess &&
        hipMemcpy(d_xy, xy, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice) == hipSuccess &&
        (!disp || (hipMalloc((void **)&d_dsp, sizeof(int32_t) * n) == hipSuccess && hipMemcpy(d_dsp, disp, sizeof(int32_t) * n, hipMemcpyHostToDevice) == hipSuccess))) {
        if (launch_delaunay_prepare_test(d_xy, d_dsp, n, d_ord, width, height, step, disp_max, nullptr) != 0) {
            rc = SV_ERR_UNSUPPORTED;
        } else if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(ord.data(), d_ord, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyDeviceToHost) == hipSuccess) {
            rc = ord[0];
            if (rc > 0) memcpy(ids_out, ord.data() + 1, sizeof(int32_t) * (size_t)rc);
        }
    }
    if (d_xy) (void)hipFree(d_xy);
    if (d_ord) (void)hipFree(d_ord);
    if (d_dsp) (void)hipFree(d_dsp);
    return rc;
}

// Test hook: the adaptive mean divides by v_rcp_f32 + one FMA correction instead of the IEEE sequence; this compares the two for all
// 2^23 mantissas x both signs x 31 exponents x the sixteen possible divisors on the current device.  Returns the number of
// differing quotients (0 expected; *first_a / *first_d = bits of the first one), or < 0.  *control: the same count for a * rcp(d)
// without the correction (> 0: the comparison does discriminate).
long long sv_debug_check_amean_div(unsigned int *first_a, unsigned int *first_d, long long *control) {
    unsigned long long *d = nullptr, hres[4] = {0, 0, 0, 0};
    if (hipMalloc((void **)&d, sizeof(hres)) != hipSuccess) return SV_ERR_HIP;
    long long rc = SV_ERR_HIP;
    if (hipMemset(d, 0, sizeof(hres)) == hipSuccess && launch_check_amean_div(d, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(hres, d, sizeof(hres), hipMemcpyDeviceToHost) == hipSuccess) {
        rc = (long long)hres[0];
        if (first_a) *first_a = (unsigned int)hres[1];
        if (first_d) *first_d = (unsigned int)hres[2];
        if (control) *control = (long long)hres[3];
    }
    (void)hipFree(d);
    return rc;
}

int sv_host_delaunay_split(const int32_t *xy, int n, int32_t *tri_out, int cap, int helper_delay_us) {
    return sv_host_delaunay_par(xy, n, tri_out, cap, 1, helper_delay_us);
}

int sv_host_delaunay_par(const int32_t *xy, int n, int32_t *tri_out, int cap, int depth, int helper_delay_us) {
    if (!xy || !tri_out || depth < 1 || depth > 8) return SV_ERR_ARG;
    struct Helper {
        std::vector<std::thread> threads;
        std::mutex mu;  // at depth > 1 helpers spawn helpers
        int delay_us;
        static void run(void *ctx, void (*fn)(void *), void *arg) {
            Helper *hp = static_cast<Helper *>(ctx);
            const int delay = hp->delay_us;
            std::lock_guard<std::mutex> lk(hp->mu);
            hp->threads.emplace_back([fn, arg, delay] {
                if (delay > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay));
                fn(arg);
            });
        }
    } helper{{}, {}, helper_delay_us};
    const Delaunay::Spawn spawn{&Helper::run, &helper, depth};
    Delaunay dl;
    const int nt = dl.triangulate(xy, n, tri_out, cap, &spawn);
    for (size_t i = 0;; i++) {  // a late helper finds the work claimed and returns at once (joined by index: the vector may still grow)
        std::thread t;
        {
            std::lock_guard<std::mutex> lk(helper.mu);
            if (i >= helper.threads.size()) break;
            t = std::move(helper.threads[i]);
        }
        t.join();
    }
    return nt;
}

} /* extern "C" */
