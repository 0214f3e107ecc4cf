// Batch engine + C ABI (group B of include/stereo_vision_hip.h).
//
// Execution model (MI355X-first, not the reference's one-pair-per-call globals):
//   * one sv_handle = one GPU; S "driver" threads, each owning one HIP stream and one slot of device buffers for
//     `chunk` pairs, plus a pool of W host worker threads shared by all drivers;
//   * a batch is cut into chunks that drivers pull from an atomic counter.  Per chunk a driver runs
//       phase 1 (GPU)  descriptors + support matching            -> D2H of the small support lattices
//       host stage     in-place lattice filters + 2 Delaunay triangulations per pair (order-dependent /
//                      pointer-chasing work that the reference also does on the CPU), fanned out over the pool,
//                      results packed into one pinned blob by atomic bump allocation
//       phase 2 (GPU)  one H2D of the blob, then plane fit + raster, grid, dense matching, L/R check, speckle,
//                      gap interpolation, adaptive mean, median, output
//     so that while one driver waits for the pool the other drivers' (large) launches keep the GPU busy.
//   * no allocation, no hipMalloc and no device-wide synchronisation inside the per-batch path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/stereo_vision_hip.h"
#include "host_stage.h"
#include "sv_kernels.h"

using namespace sv;

namespace {

thread_local std::string g_create_error;

struct TimedLaunch {
    int id;
    hipEvent_t a, b;
};

struct HostScratch {  // per pool thread
    Delaunay dl;
    std::vector<int32_t> xy;
    std::vector<int32_t> sup;
};

struct Worker {  // a "driver": one stream + one slot of device buffers
    sv_handle *h = nullptr;
    int id = 0;
    std::thread th;
    hipStream_t stream = nullptr;
    SlotDev dev{};
    // pinned host staging
    int16_t *h_dcan = nullptr;      // [cap][Hc*Wc]
    int32_t *h_blob[2] = {nullptr, nullptr};
    hipEvent_t blob_copied[2] = {nullptr, nullptr};
    bool blob_pending[2] = {false, false};
    int parity = 0;
    size_t blob_words = 0;
    // host-stage fan-out state of the chunk in flight
    int32_t *cur_blob = nullptr;
    int cur_i0 = 0;
    std::atomic<size_t> blob_off{0};
    std::atomic<int> pending{0};
    std::mutex pmu;
    std::condition_variable pcv;
    std::string task_error;
    // timing
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
};

struct Task {
    Worker *w;
    int pair;  // index inside the chunk
    int side;  // -1: filter stage (then triangulates side 0 and queues side 1); 1: triangulation of the right image
};

struct Job {
    const uint8_t *left = nullptr, *right = nullptr;
    int batch = 0, stride = 0;
    float *d1 = nullptr, *d2 = nullptr;
    int32_t *status = nullptr;
    std::atomic<int> next{0};
};

}  // namespace

struct sv_handle {
    sv_params p;
    sv_config cfg;
    KParams kp;
    int nproc = 1;  // maps per pair that get post-processed
    int chunk = 1;
    std::vector<Worker *> workers;  // drivers
    // host pool
    std::vector<std::thread> pool;
    std::vector<HostScratch *> scratch;
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<Task> queue;
    bool pool_quit = false;
    // job control
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    Job job;
    uint64_t generation = 0;
    int running = 0;
    bool quit = false;
    std::string error;
    bool failed = false;
    // timing
    bool timing = false;
    std::mutex tmu;
    double k_ms[K_COUNT] = {0};
    int64_t k_calls[K_COUNT] = {0};
    // debug
    std::map<std::string, std::vector<uint8_t>> dbg;
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

int validate(const sv_params &p, const sv_config &c, std::string &err) {
    char b[256];
    if (p.subsampling) {
        err = "subsampling (half-resolution mode) is not supported";
        return SV_ERR_UNSUPPORTED;
    }
    if (p.disp_min != 0) {
        err = "disp_min must be 0";
        return SV_ERR_UNSUPPORTED;
    }
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
    d.Wc = (d.W + d.step - 1) / d.step;
    d.Hc = (d.H + d.step - 1) / d.step;
    d.grid_size = p.grid_size;
    d.gw = (int)ceil((float)d.W / (float)p.grid_size);  // elas.cpp:88-89
    d.gh = (int)ceil((float)d.H / (float)p.grid_size);
    d.ncell = d.gw * d.gh;
    d.disp_max = p.disp_max;
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
    h->nproc = p.postprocess_only_left ? 1 : 2;
}

void timing_hook(void *ctx, int id, bool before, hipStream_t st) {
    Worker *w = (Worker *)ctx;
    if (before) {
        if (w->events_used + 2 > w->event_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            w->event_pool.push_back(a);
            w->event_pool.push_back(b);
        }
        TimedLaunch t{id, w->event_pool[w->events_used], w->event_pool[w->events_used + 1]};
        w->events_used += 2;
        w->timed.push_back(t);
        hipEventRecord(t.a, st);
    } else if (!w->timed.empty()) {
        hipEventRecord(w->timed.back().b, st);
    }
}

void collect_timing(Worker *w) {
    sv_handle *h = w->h;
    if (w->timed.empty()) return;
    std::lock_guard<std::mutex> g(h->tmu);
    for (const TimedLaunch &t : w->timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            h->k_ms[t.id] += ms;
            h->k_calls[t.id] += 1;
        }
    }
    w->timed.clear();
    w->events_used = 0;
}

template <class T>
void dbg_put(sv_handle *h, const char *name, const T *data, size_t count) {
    std::vector<uint8_t> &v = h->dbg[name];
    v.resize(count * sizeof(T));
    if (count) memcpy(v.data(), data, count * sizeof(T));
}

void dbg_from_device(sv_handle *h, Worker *w, const char *name, const void *dptr, size_t bytes) {
    std::vector<uint8_t> &v = h->dbg[name];
    v.resize(bytes);
    HIP_TRY(hipStreamSynchronize(w->stream));
    HIP_TRY(hipMemcpy(v.data(), dptr, bytes, hipMemcpyDeviceToHost));
}

void dbg_maps(sv_handle *h, Worker *w, const char *stage, const float *base, int j) {
    const size_t N = h->kp.d.N;
    char name[64];
    for (int side = 0; side < 2; side++) {
        snprintf(name, sizeof(name), "%s%d", stage, side + 1);
        dbg_from_device(h, w, name, base + ((size_t)j * 2 + side) * N, N * sizeof(float));
    }
}

// compacted candidate lists in the reference's layout (elas.cpp:631-648) from the device bit masks
void dbg_grid(sv_handle *h, Worker *w, int j) {
    const Dims &d = h->kp.d;
    std::vector<uint32_t> m((size_t)d.ncell * d.MW);
    for (int side = 0; side < 2; side++) {
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(hipMemcpy(m.data(), w->dev.gmaskB + ((size_t)j * 2 + side) * d.ncell * d.MW, m.size() * 4, hipMemcpyDeviceToHost));
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

void process_chunk(Worker *w, int i0, int n) {
    sv_handle *h = w->h;
    const KParams &k = h->kp;
    const Dims &d = k.d;
    const Job &job = h->job;
    const bool dbg = h->cfg.keep_debug != 0;
    hipStream_t st = w->stream;
    const size_t in_pair = (size_t)d.H * job.stride;
    const int lat = d.Wc * d.Hc;

    // ---- phase 1
    launch_descriptor(k, job.left + (size_t)i0 * in_pair, job.right + (size_t)i0 * in_pair, in_pair, job.stride, w->dev, n, st);
    launch_support(k, w->dev, n, st);
    HIP_TRY(hipMemcpyAsync(w->h_dcan, w->dev.dcan, sizeof(int16_t) * (size_t)n * lat, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (dbg) {
        const int j = n - 1;
        dbg_from_device(h, w, "desc1", w->dev.desc + ((size_t)j * 2) * d.N * 16, (size_t)d.N * 16);
        dbg_from_device(h, w, "desc2", w->dev.desc + ((size_t)j * 2 + 1) * d.N * 16, (size_t)d.N * 16);
        dbg_put(h, "dcan_raw", w->h_dcan + (size_t)j * lat, lat);
        int32_t dd[2] = {d.Wc, d.Hc};
        dbg_put(h, "dcan_dims", dd, 2);
    }

    // ---- host stage: filters + Delaunay fanned out over the pool, packed into one blob
    const int b = w->parity;
    w->parity ^= 1;
    if (w->blob_pending[b]) {
        HIP_TRY(hipEventSynchronize(w->blob_copied[b]));
        w->blob_pending[b] = false;
    }
    int32_t *blob = w->h_blob[b];
    w->cur_blob = blob;
    w->cur_i0 = i0;
    w->blob_off.store((size_t)w->dev.cap * META_WORDS);
    w->task_error.clear();
    w->pending.store(n);
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        for (int j = 0; j < n; j++) h->queue.push_back(Task{w, j, -1});
    }
    h->qcv.notify_all();
    {
        std::unique_lock<std::mutex> lk(w->pmu);
        w->pcv.wait(lk, [&] { return w->pending.load() == 0; });
    }
    if (!w->task_error.empty()) throw std::runtime_error(w->task_error);
    const size_t off = w->blob_off.load();
    if (dbg) {
        const int32_t *meta = blob + (size_t)(n - 1) * META_WORDS;
        if (meta[0] >= 3) {
            dbg_put(h, "support", blob + meta[1], (size_t)meta[0] * 3);
            dbg_put(h, "tri1", blob + meta[3], (size_t)meta[2] * 3);
            dbg_put(h, "tri2", blob + meta[5], (size_t)meta[4] * 3);
        } else {
            dbg_put(h, "support", w->h_dcan, 0);
        }
    }

    // ---- phase 2
    HIP_TRY(hipMemcpyAsync(w->dev.blob, blob, off * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(w->blob_copied[b], st));
    w->blob_pending[b] = true;
    launch_grid(k, w->dev, n, st);
    launch_triangles(k, w->dev, n, st);
    launch_dense(k, w->dev, n, st);
    launch_lr(k, w->dev, n, st);
    const bool active = dbg && blob[(size_t)(n - 1) * META_WORDS] >= 3;
    if (active) {
        const int j = n - 1;
        const int32_t *meta = blob + (size_t)j * META_WORDS;
        dbg_grid(h, w, j);
        dbg_from_device(h, w, "planes1", w->dev.planes + ((size_t)j * 2) * d.max_tri * 6, (size_t)meta[2] * 6 * sizeof(float));
        dbg_from_device(h, w, "planes2", w->dev.planes + ((size_t)j * 2 + 1) * d.max_tri * 6, (size_t)meta[4] * 6 * sizeof(float));
        dbg_from_device(h, w, "tri_id1", w->dev.tri_id + ((size_t)j * 2) * d.N, (size_t)d.N * 4);
        dbg_from_device(h, w, "tri_id2", w->dev.tri_id + ((size_t)j * 2 + 1) * d.N, (size_t)d.N * 4);
        dbg_maps(h, w, "wta", w->dev.wta, j);
        dbg_maps(h, w, "lr", w->dev.disp, j);
    }
    launch_speckle(k, w->dev, n, h->nproc, st);
    if (active) dbg_maps(h, w, "speckle", w->dev.disp, n - 1);
    launch_gap_rows(k, w->dev, n, h->nproc, st);
    launch_gap_cols(k, w->dev, n, h->nproc, st);
    if (active) dbg_maps(h, w, "gap", w->dev.disp, n - 1);
    if (h->p.filter_adaptive_mean) launch_amean(k, w->dev, n, h->nproc, st);
    if (active) dbg_maps(h, w, "amean", w->dev.disp, n - 1);
    if (h->p.filter_median) launch_median(k, w->dev, n, h->nproc, st);
    if (active) dbg_maps(h, w, "final", w->dev.disp, n - 1);
    launch_output(k, w->dev, n, job.d1 + (size_t)i0 * d.N, job.d2 ? job.d2 + (size_t)i0 * d.N : nullptr, st);
    HIP_TRY(hipGetLastError());
}


void pair_done(Worker *w) {
    if (w->pending.fetch_sub(1) == 1) {
        std::lock_guard<std::mutex> lk(w->pmu);
        w->pcv.notify_all();
    }
}

void note_error(Worker *w, const char *what) {
    std::lock_guard<std::mutex> lk(w->pmu);
    if (w->task_error.empty()) w->task_error = what;
}

// one Delaunay triangulation of a pair's support points; errors are recorded, never thrown (the caller's
// completion accounting must run in any case)
void triangulate_side(sv_handle *h, HostScratch *sc, Worker *w, int j, int side) {
    const Dims &d = h->kp.d;
    int32_t *blob = w->cur_blob;
    int32_t *meta = blob + (size_t)j * META_WORDS;
    const int ns = meta[0];
    const int32_t *sup = blob + meta[1];
    if ((int)sc->xy.size() < 2 * ns) sc->xy.resize(2 * ns);
    for (int q = 0; q < ns; q++) {  // elas.cpp:449-461: left uses (u,v), right (u-d,v)
        sc->xy[2 * q] = side ? sup[3 * q] - sup[3 * q + 2] : sup[3 * q];
        sc->xy[2 * q + 1] = sup[3 * q + 1];
    }
    const int nt = sc->dl.triangulate(sc->xy.data(), ns, blob + meta[3 + 2 * side], 2 * ns);
    if (nt < 0 || nt > d.max_tri) {
        note_error(w, "triangle capacity exceeded");
        meta[2 + 2 * side] = 0;
        return;
    }
    meta[2 + 2 * side] = nt;
}

void run_task(sv_handle *h, HostScratch *sc, const Task &t) {
    Worker *w = t.w;
    const Dims &d = h->kp.d;
    const int lat = d.Wc * d.Hc;
    int32_t *blob = w->cur_blob;
    int32_t *meta = blob + (size_t)t.pair * META_WORDS;
    if (t.side >= 0) {
        triangulate_side(h, sc, w, t.pair, t.side);
        if (__atomic_add_fetch(&meta[6], 1, __ATOMIC_ACQ_REL) == 2) pair_done(w);
        return;
    }
    if ((int)sc->sup.size() < d.max_pts * 3) sc->sup.resize((size_t)d.max_pts * 3);
    int ns = support_filter(h->p, w->h_dcan + (size_t)t.pair * lat, d.W, d.H, sc->sup.data(), d.max_pts);
    if (ns < 0) {
        note_error(w, "support point capacity exceeded");
        ns = 0;
    }
    if (h->job.status) h->job.status[w->cur_i0 + t.pair] = ns;
    meta[0] = ns;
    meta[1] = meta[3] = meta[5] = 0;
    meta[2] = meta[4] = 0;
    meta[6] = meta[7] = 0;
    if (ns < 3) {  // elas.cpp:63-69
        pair_done(w);
        return;
    }
    // 3*ns words of points + two triangle lists of at most 2*ns triangles each
    const size_t need = (size_t)ns * 3 + 2 * ((size_t)2 * ns * 3);
    const size_t off = w->blob_off.fetch_add(need);
    if (off + need > w->blob_words) {
        note_error(w, "host blob overflow");
        meta[0] = 0;
        pair_done(w);
        return;
    }
    meta[1] = (int32_t)off;
    meta[3] = (int32_t)(off + (size_t)ns * 3);
    meta[5] = (int32_t)(off + (size_t)ns * 3 + (size_t)2 * ns * 3);
    memcpy(blob + off, sc->sup.data(), sizeof(int32_t) * (size_t)ns * 3);
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        h->queue.push_front(Task{w, t.pair, 1});
    }
    h->qcv.notify_one();
    triangulate_side(h, sc, w, t.pair, 0);
    if (__atomic_add_fetch(&meta[6], 1, __ATOMIC_ACQ_REL) == 2) pair_done(w);
}

void pool_main(sv_handle *h, HostScratch *sc) {
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(h->qmu);
            h->qcv.wait(lk, [&] { return h->pool_quit || !h->queue.empty(); });
            if (h->pool_quit && h->queue.empty()) return;
            t = h->queue.front();
            h->queue.pop_front();
        }
        run_task(h, sc, t);
    }
}

void worker_main(Worker *w) {
    sv_handle *h = w->h;
    hipSetDevice(h->cfg.device);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(h->mu);
            h->cv_start.wait(lk, [&] { return h->quit || h->generation != seen; });
            if (h->quit) return;
            seen = h->generation;
        }
        g_launch_hook.fn = h->timing ? timing_hook : nullptr;
        g_launch_hook.ctx = w;
        try {
            for (;;) {
                const int i0 = h->job.next.fetch_add(h->chunk);
                if (i0 >= h->job.batch) break;
                process_chunk(w, i0, std::min(h->chunk, h->job.batch - i0));
            }
            HIP_TRY(hipStreamSynchronize(w->stream));
            if (h->timing) collect_timing(w);
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(h->mu);
            h->failed = true;
            h->error = e.what();
            h->job.next.store(h->job.batch);
            hipStreamSynchronize(w->stream);
        }
        {
            std::lock_guard<std::mutex> lk(h->mu);
            if (--h->running == 0) h->cv_done.notify_all();
        }
    }
}

template <class T>
void dev_alloc(T *&p, size_t count) {
    HIP_TRY(hipMalloc((void **)&p, count * sizeof(T)));
}

void alloc_worker(sv_handle *h, Worker *w) {
    const Dims &d = h->kp.d;
    const size_t cap = (size_t)h->chunk;
    SlotDev &s = w->dev;
    s.cap = (int)cap;
    HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    dev_alloc(s.desc, cap * 2 * d.N * 16);
    dev_alloc(s.dcan, cap * d.Wc * d.Hc);
    w->blob_words = cap * (META_WORDS + (size_t)d.max_pts * 3 + 2 * (size_t)d.max_tri * 3 + 64);
    dev_alloc(s.blob, w->blob_words);
    dev_alloc(s.trirec, cap * 2 * d.max_tri);
    {
        uint8_t *r = nullptr;
        dev_alloc(r, cap * 2 * (size_t)d.max_tri * 36);
        s.rrec = r;
    }
    dev_alloc(s.planes, cap * 2 * d.max_tri * 6);
    dev_alloc(s.gmaskA, cap * 2 * d.ncell * d.MW);
    dev_alloc(s.gmaskB, cap * 2 * d.ncell * d.MW);
    dev_alloc(s.tri_id, cap * 2 * d.N);
    dev_alloc(s.wta, cap * 2 * d.N);
    dev_alloc(s.disp, cap * 2 * d.N);
    dev_alloc(s.tmp, cap * 2 * d.N);
    dev_alloc(s.csize, cap * 2 * d.N);
    HIP_TRY(hipHostMalloc((void **)&w->h_dcan, sizeof(int16_t) * cap * d.Wc * d.Hc, hipHostMallocDefault));
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipHostMalloc((void **)&w->h_blob[b], sizeof(int32_t) * w->blob_words, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&w->blob_copied[b], hipEventDisableTiming));
    }
}

void free_worker(Worker *w) {
    SlotDev &s = w->dev;
    void *dptrs[] = {s.desc, s.dcan, s.blob, s.rrec, s.trirec, s.planes, s.gmaskA, s.gmaskB, s.tri_id, s.wta, s.disp, s.tmp, s.csize};
    for (void *p : dptrs)
        if (p) hipFree(p);
    if (w->h_dcan) hipHostFree(w->h_dcan);
    for (int b = 0; b < 2; b++) {
        if (w->h_blob[b]) hipHostFree(w->h_blob[b]);
        if (w->blob_copied[b]) hipEventDestroy(w->blob_copied[b]);
    }
    for (hipEvent_t e : w->event_pool) hipEventDestroy(e);
    if (w->stream) hipStreamDestroy(w->stream);
}

int run_job(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    if (!h) return SV_ERR_ARG;
    if (!left || !right || !d1 || batch < 0 || stride < h->cfg.width) {
        h->error = "bad argument (null pointer, negative batch, or stride < width)";
        return SV_ERR_ARG;
    }
    if (batch == 0) return SV_OK;
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->running != 0) {
        h->error = "a batch is already in flight on this handle";
        return SV_ERR_STATE;
    }
    h->job.left = left;
    h->job.right = right;
    h->job.batch = batch;
    h->job.stride = stride;
    h->job.d1 = d1;
    h->job.d2 = d2;
    h->job.status = status;
    h->job.next.store(0);
    h->failed = false;
    h->running = (int)h->workers.size();
    h->generation++;
    h->cv_start.notify_all();
    h->cv_done.wait(lk, [&] { return h->running == 0; });
    return h->failed ? SV_ERR_HIP : SV_OK;
}

}  // namespace

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
    h->p = *params;
    h->cfg = *cfg;
    fill_kparams(h);
    int npool = cfg->n_workers > 0 ? cfg->n_workers : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    int nw = cfg->n_streams > 0 ? cfg->n_streams : 4;
    h->chunk = cfg->chunk > 0 ? cfg->chunk : 16;
    if (cfg->keep_debug) {
        nw = 1;
        h->chunk = 1;
    }
    try {
        HIP_TRY(hipSetDevice(cfg->device));
        for (int i = 0; i < nw; i++) {
            Worker *w = new Worker();
            w->h = h;
            w->id = i;
            h->workers.push_back(w);
            alloc_worker(h, w);
        }
    } catch (const std::exception &e) {
        g_create_error = e.what();
        for (Worker *w : h->workers) {
            free_worker(w);
            delete w;
        }
        delete h;
        return SV_ERR_HIP;
    }
    for (int i = 0; i < npool; i++) {
        HostScratch *sc = new HostScratch();
        h->scratch.push_back(sc);
        h->pool.emplace_back(pool_main, h, sc);
    }
    for (Worker *w : h->workers) w->th = std::thread(worker_main, w);
    *out = h;
    return SV_OK;
}

int sv_destroy(sv_handle *h) {
    if (!h) return SV_ERR_ARG;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        h->quit = true;
        h->cv_start.notify_all();
    }
    for (Worker *w : h->workers)
        if (w->th.joinable()) w->th.join();
    {
        std::lock_guard<std::mutex> lk(h->qmu);
        h->pool_quit = true;
        h->qcv.notify_all();
    }
    for (std::thread &t : h->pool)
        if (t.joinable()) t.join();
    for (HostScratch *sc : h->scratch) delete sc;
    (void)hipSetDevice(h->cfg.device);
    for (Worker *w : h->workers) {
        free_worker(w);
        delete w;
    }
    delete h;
    return SV_OK;
}

const char *sv_last_error(const sv_handle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int sv_process_batch_device(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    return run_job(h, left, right, batch, stride, d1, d2, status);
}

int sv_process_batch_host(sv_handle *h, const uint8_t *left, const uint8_t *right, int batch, int stride, float *d1, float *d2, int32_t *status) {
    if (!h) return SV_ERR_ARG;
    if (!left || !right || !d1 || batch < 0 || stride < h->cfg.width) {
        h->error = "bad argument";
        return SV_ERR_ARG;
    }
    if (batch == 0) return SV_OK;
    const Dims &d = h->kp.d;
    const size_t in_bytes = (size_t)batch * d.H * stride, out_bytes = (size_t)batch * d.N * sizeof(float);
    uint8_t *dl = nullptr, *dr = nullptr;
    float *o1 = nullptr, *o2 = nullptr;
    int rc = SV_OK;
    try {
        HIP_TRY(hipSetDevice(h->cfg.device));
        HIP_TRY(hipMalloc((void **)&dl, in_bytes));
        HIP_TRY(hipMalloc((void **)&dr, in_bytes));
        HIP_TRY(hipMalloc((void **)&o1, out_bytes));
        if (d2) HIP_TRY(hipMalloc((void **)&o2, out_bytes));
        HIP_TRY(hipMemcpy(dl, left, in_bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dr, right, in_bytes, hipMemcpyHostToDevice));
        // pairs the reference would leave untouched (<3 support points) must keep the caller's values
        HIP_TRY(hipMemcpy(o1, d1, out_bytes, hipMemcpyHostToDevice));
        if (d2) HIP_TRY(hipMemcpy(o2, d2, out_bytes, hipMemcpyHostToDevice));
        rc = run_job(h, dl, dr, batch, stride, o1, o2, status);
        if (rc == SV_OK) {
            HIP_TRY(hipMemcpy(d1, o1, out_bytes, hipMemcpyDeviceToHost));
            if (d2) HIP_TRY(hipMemcpy(d2, o2, out_bytes, hipMemcpyDeviceToHost));
        }
    } catch (const std::exception &e) {
        h->error = e.what();
        rc = SV_ERR_HIP;
    }
    if (dl) hipFree(dl);
    if (dr) hipFree(dr);
    if (o1) hipFree(o1);
    if (o2) hipFree(o2);
    return rc;
}

int sv_elas_process(sv_handle *h, const uint8_t *I1, const uint8_t *I2, float *D1, float *D2, const int32_t *dims) {
    if (!h || !dims) return SV_ERR_ARG;
    if (dims[0] != h->cfg.width || dims[1] != h->cfg.height) {
        h->error = "dims do not match the handle's width/height";
        return SV_ERR_ARG;
    }
    return sv_process_batch_host(h, I1, I2, 1, dims[2], D1, D2, nullptr);
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
    return K_COUNT;
}

void sv_kernel_times_reset(sv_handle *h) {
    if (!h) return;
    std::lock_guard<std::mutex> g(h->tmu);
    for (int i = 0; i < K_COUNT; i++) {
        h->k_ms[i] = 0;
        h->k_calls[i] = 0;
    }
}

void sv_kernel_timing_enable(sv_handle *h, int on) {
    if (h) h->timing = on != 0;
}

int sv_host_support_filter(const sv_params *p, int16_t *dcan, int width, int height, int32_t *support, int cap) {
    if (!p || !dcan || !support) return SV_ERR_ARG;
    return support_filter(*p, dcan, width, height, support, cap);
}

int sv_host_delaunay(const int32_t *xy, int n, int32_t *tri_out, int cap) {
    if (!xy || !tri_out) return SV_ERR_ARG;
    Delaunay dl;
    return dl.triangulate(xy, n, tri_out, cap);
}

} /* extern "C" */
