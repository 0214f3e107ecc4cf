// Group (A) of include/stereo_vision_hip.h: the reference's three exported symbols, so that the reference's
// ctypes binding (stereo_vision/sv.py:164-192) can load this library in place of stereo_vision_serial*.so.
//
//   generatePointCloud   reference: src/serial_includes/main/stereo_vision.cpp:565-623
//     first call  -> externalInit (:498-562): freeze width/height/scale/calibration, stereoRectify -> Q, allocate `points`
//     every call  -> wrap BGRA buffers, resize to the frozen size (:587-591; a copy when the sizes agree), cvtColor BGRA2GRAY (:338-339),
//                    generateDisparityMap (:296-318: Elas MIDDLEBURY + postprocess_only_left + adaptive mean, then
//                    convertTo(CV_8UC1, 4.0)), publishPointCloud (:222-259: (X,Y,Z) = Q*[x y d 1] / w)
//   clean / getColor     :105-114 / :625-627
//
// Everything per frame runs on the GPU: gray conversion, the ELAS engine, the x4 u8 conversion and the reprojection
// (legacy_kernels.hip); the host only moves the caller's buffers in and the point array out.
//
// Documented deviations from the reference: clean() does not exit(0); `points` is filled on every call
// (the reference only fills it when graphics==true and otherwise returns uninitialised memory); YOLO object tracking,
// the GLUT viewer and imshow windows are not part of this library (objectTracking/graphics/display are accepted and
// ignored).  Arguments 15 and 16 (removeSky, subsampling) are NOT read: the reference's own binding passes 14 arguments
// (stereo_vision/sv.py:180,189), so a callee that read them would read whatever the caller's registers / stack hold.  Half-
// resolution mode is selected with sv_legacy_set_subsampling(1) before the first frame instead (our sv.py does that).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/stereo_vision_hip.h"
#include "calib.h"

// engine.cpp (not part of the public header): one device-resident pair on a latency handle whose first GPU phase waits for `ready`
// on the device instead of the caller synchronising its stream on the host
int sv_internal_process_after(sv_handle *h, hipEvent_t ready, const uint8_t *left, const uint8_t *right, int stride, float *d1, float *d2);
int sv_internal_copy2(sv_handle *h, void *dst_a, const void *src_a, void *dst_b, const void *src_b, size_t bytes_each);

namespace sv {
void launch_resize_bgra(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh, hipStream_t st);
void launch_bgra_to_gray(const unsigned char *bgra_l, const unsigned char *bgra_r, unsigned char *gray_l, unsigned char *gray_r, int n, hipStream_t st);
void launch_dmap_and_cloud(const float *disp, unsigned char *dmap, double *points, const double *Q16, int W, int H, hipStream_t st);
void launch_remap_gray(const unsigned char *src, unsigned char *dst, const float *mapx, const float *mapy, int W, int H, hipStream_t st);
int launch_disp_to_u8(const float *disp, size_t n, unsigned char *out, hipStream_t st);
int launch_reproject_batch(const float *disp, int batch, int W, int H, const double *Q16, const double *XR9, const double *XT3, unsigned char *dmap, double *points,
                           hipStream_t st);
}  // namespace sv

namespace {

struct Legacy {
    bool ready = false;
    bool failed = false;
    int W = 0, H = 0;
    sv_handle *engine = nullptr;
    sv::Rectification rect;
    double *d_Q = nullptr;
    unsigned char *d_bgra_l = nullptr, *d_bgra_r = nullptr, *d_gray_l = nullptr, *d_gray_r = nullptr, *d_dmap = nullptr;
    float *d_disp = nullptr, *d_disp2 = nullptr;
    double *d_points = nullptr;
    // rectification remap (stereo_vision.cpp:341, commented out in the reference; here: off unless sv_legacy_set_rectify(1))
    bool rectify = false;
    float *d_maps = nullptr;            // [4][H][W]: lmapx, lmapy, rmapx, rmapy (stereo_vision.cpp:477-478)
    unsigned char *d_rect_l = nullptr, *d_rect_r = nullptr;
    std::vector<float> h_maps;
    Double3 *points = nullptr;          // host, library-owned (stereo_vision.cpp:89-93)
    std::vector<Uchar4> colors;         // last left image
    unsigned char *h_dmap = nullptr;    // last u8 disparity image (host, page-locked)
    hipStream_t stream = nullptr;
    int device = 0;
    hipEvent_t ev_in = nullptr;         // inputs of the matcher are ready (the engine's first stream waits for it: no host sync)
    // upload staging: the caller's buffers are pageable and only valid during the call - they are copied into page-locked
    // memory piece by piece, each piece's DMA running while the next is being copied
    unsigned char *h_stage = nullptr;
    size_t stage_bytes = 0;
    unsigned char *d_src_l = nullptr, *d_src_r = nullptr;  // frames of another size than the frozen one, before the resize
    size_t src_bytes = 0;
};

Legacy g;
std::mutex g_mu;
bool g_want_rectify = false;  // sv_legacy_set_rectify: read by the first generatePointCloud call
bool g_want_subsampling = false;  // sv_legacy_set_subsampling: likewise
int g_want_device = 0;            // sv_legacy_set_device: likewise

#define L_TRY(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "stereo_vision_hip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            return false;                                                                        \
        }                                                                                        \
    } while (0)

bool legacy_init(int width, int height, float scale, const char *yaml, bool subsampling) {
    g.W = width;
    g.H = height;
    sv::Calibration c;
    std::string err;
    printf("Using CAMERA_CALIBRATION_YAML : %s\n", yaml ? yaml : "(null)");
    if (!sv::load_calibration_yaml(yaml, c, err)) {
        fprintf(stderr, "stereo_vision_hip: %s\n", err.c_str());
        return false;
    }
    for (int i = 0; i < 6; i++) {  // K1, K2 first two rows /= scale_factor (stereo_vision.cpp:364-376)
        c.K1[i] /= scale;
        c.K2[i] /= scale;
    }
    sv::stereo_rectify(c, width, height, width, height, 0.0, g.rect);  // :439, calib_img_size == out_img_size (:524-525)

    sv_params p;
    sv_params_init(&p, SV_DRIVER);  // stereo_vision.cpp:307-311 (disp_max stays 255)
    // param.subsampling = subsample (:309).  Elas then fills only the first (W/2)*(H/2) floats of the zeroed full-size
    // leftdpf (:304, elas.h:160-161) and the driver converts / reprojects the whole W x H buffer as it stands; the engine
    // writes the same floats into the same zeroed buffer, so the u8 map and the cloud come out as the reference's do.
    p.subsampling = subsampling ? 1 : 0;
    sv_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.width = width;
    cfg.height = height;
    cfg.device = g.device = g_want_device;
    cfg.n_workers = 7;  // latency mode: the frame's host stage shared with up to seven polling threads next to the calling one (sv_config.latency_split)
    cfg.n_streams = 1;
    cfg.n_slots = 2;
    cfg.chunk = 1;
    if (sv_create(&p, &cfg, &g.engine) != SV_OK) {
        fprintf(stderr, "stereo_vision_hip: %s\n", sv_last_error(nullptr));
        return false;
    }
    const size_t N = (size_t)width * height;
    L_TRY(hipSetDevice(g.device));
    L_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    L_TRY(hipEventCreateWithFlags(&g.ev_in, hipEventDisableTiming));
    g.stage_bytes = 2 * N * 4;
    L_TRY(hipHostMalloc((void **)&g.h_stage, g.stage_bytes, hipHostMallocDefault));
    L_TRY(hipMalloc((void **)&g.d_bgra_l, N * 4));
    L_TRY(hipMalloc((void **)&g.d_bgra_r, N * 4));
    L_TRY(hipMalloc((void **)&g.d_gray_l, N));
    L_TRY(hipMalloc((void **)&g.d_gray_r, N));
    L_TRY(hipMalloc((void **)&g.d_dmap, N));
    L_TRY(hipMalloc((void **)&g.d_disp, N * sizeof(float)));
    L_TRY(hipMalloc((void **)&g.d_disp2, N * sizeof(float)));
    L_TRY(hipMalloc((void **)&g.d_points, N * 3 * sizeof(double)));
    L_TRY(hipMalloc((void **)&g.d_Q, 16 * sizeof(double)));
    L_TRY(hipMemcpy(g.d_Q, g.rect.Q, 16 * sizeof(double), hipMemcpyHostToDevice));
    g.rectify = g_want_rectify;
    if (g.rectify) {  // findRectificationMap's two initUndistortRectifyMap calls (stereo_vision.cpp:477-478)
        g.h_maps.assign(4 * N, 0.f);
        if (!sv::init_undistort_rectify_map(c.K1, c.D1, g.rect.R1, g.rect.P1, width, height, g.h_maps.data(), g.h_maps.data() + N) ||
            !sv::init_undistort_rectify_map(c.K2, c.D2, g.rect.R2, g.rect.P2, width, height, g.h_maps.data() + 2 * N, g.h_maps.data() + 3 * N)) {
            fprintf(stderr, "stereo_vision_hip: singular rectification\n");
            return false;
        }
        L_TRY(hipMalloc((void **)&g.d_maps, 4 * N * sizeof(float)));
        L_TRY(hipMemcpy(g.d_maps, g.h_maps.data(), 4 * N * sizeof(float), hipMemcpyHostToDevice));
        L_TRY(hipMalloc((void **)&g.d_rect_l, N));
        L_TRY(hipMalloc((void **)&g.d_rect_r, N));
    }
    // page-locked: the 11 MB of points per frame come back at PCIe speed instead of through a pageable bounce buffer
    L_TRY(hipHostMalloc((void **)&g.points, N * sizeof(Double3), hipHostMallocDefault));
    memset(g.points, 0, N * sizeof(Double3));
    L_TRY(hipHostMalloc((void **)&g.h_dmap, N, hipHostMallocDefault));
    memset(g.h_dmap, 0, N);
    g.colors.assign(N, Uchar4{0, 0, 0, 0});
    printf("Init done\n");
    return g.points != nullptr;
}

// host -> device through the page-locked staging buffer, in pieces: piece k+1 is copied while piece k's DMA runs
bool upload_staged(unsigned char *dev, const unsigned char *src, size_t bytes, unsigned char *stage) {
    const size_t piece = (size_t)512 << 10;
    for (size_t o = 0; o < bytes; o += piece) {
        const size_t n = bytes - o < piece ? bytes - o : piece;
        memcpy(stage + o, src + o, n);
        L_TRY(hipMemcpyAsync(dev + o, stage + o, n, hipMemcpyHostToDevice, g.stream));
    }
    return true;
}

bool legacy_frame(const unsigned char *left, const unsigned char *right, int width, int height) {
    const size_t N = (size_t)g.W * g.H, Nin = (size_t)width * height;
    const bool resize = width != g.W || height != g.H;
    L_TRY(hipSetDevice(g.device));
    if (2 * Nin * 4 > g.stage_bytes) {  // a frame larger than the frozen size: grow the staging (rare: once per new size)
        L_TRY(hipStreamSynchronize(g.stream));
        if (g.h_stage) (void)hipHostFree(g.h_stage);
        g.h_stage = nullptr;
        g.stage_bytes = 2 * Nin * 4;
        L_TRY(hipHostMalloc((void **)&g.h_stage, g.stage_bytes, hipHostMallocDefault));
    }
    if (resize && Nin * 4 > g.src_bytes) {
        L_TRY(hipStreamSynchronize(g.stream));
        if (g.d_src_l) (void)hipFree(g.d_src_l);
        if (g.d_src_r) (void)hipFree(g.d_src_r);
        g.d_src_l = g.d_src_r = nullptr;
        g.src_bytes = Nin * 4;
        L_TRY(hipMalloc((void **)&g.d_src_l, g.src_bytes));
        L_TRY(hipMalloc((void **)&g.d_src_r, g.src_bytes));
    }
    // leftdpf / rightdpf start as zeros every frame (stereo_vision.cpp:304-305)
    L_TRY(hipMemsetAsync(g.d_disp, 0, N * sizeof(float), g.stream));
    L_TRY(hipMemsetAsync(g.d_disp2, 0, N * sizeof(float), g.stream));
    // both images into the page-locked staging buffer - shared with the engine's pool threads (one thread: 120 - 150 us for 3.7 MB) -, then two DMAs
    if (sv_internal_copy2(g.engine, g.h_stage, left, g.h_stage + Nin * 4, right, Nin * 4) != SV_OK) {
        if (!upload_staged(resize ? g.d_src_l : g.d_bgra_l, left, Nin * 4, g.h_stage)) return false;
        if (!upload_staged(resize ? g.d_src_r : g.d_bgra_r, right, Nin * 4, g.h_stage + Nin * 4)) return false;
    } else {
        L_TRY(hipMemcpyAsync(resize ? g.d_src_l : g.d_bgra_l, g.h_stage, Nin * 4, hipMemcpyHostToDevice, g.stream));
        L_TRY(hipMemcpyAsync(resize ? g.d_src_r : g.d_bgra_r, g.h_stage + Nin * 4, Nin * 4, hipMemcpyHostToDevice, g.stream));
    }
    if (resize) {  // resize(left_img, left_img_OLD, out_img_size) (:590-591): cv::resize, INTER_LINEAR, 8UC4
        sv::launch_resize_bgra(g.d_src_l, width, height, g.d_bgra_l, g.W, g.H, g.stream);
        sv::launch_resize_bgra(g.d_src_r, width, height, g.d_bgra_r, g.W, g.H, g.stream);
    }
    sv::launch_bgra_to_gray(g.d_bgra_l, g.d_bgra_r, g.d_gray_l, g.d_gray_r, (int)N, g.stream);
    const unsigned char *in_l = g.d_gray_l, *in_r = g.d_gray_r;
    if (g.rectify) {  // remap(tmpL, img_left, lmapx, lmapy, INTER_LINEAR); remap(tmpR, img_right, rmapx, rmapy, INTER_LINEAR) (:341)
        sv::launch_remap_gray(g.d_gray_l, g.d_rect_l, g.d_maps, g.d_maps + N, g.W, g.H, g.stream);
        sv::launch_remap_gray(g.d_gray_r, g.d_rect_r, g.d_maps + 2 * N, g.d_maps + 3 * N, g.W, g.H, g.stream);
        in_l = g.d_rect_l, in_r = g.d_rect_r;
    }
    L_TRY(hipEventRecord(g.ev_in, g.stream));
    // the colours of the left image as the viewer gets them (left_img_OLD, :590): the staged copy when the sizes agree
    if (!resize) memcpy(g.colors.data(), g.h_stage, N * 4);  // (while the GPU converts and matches)
    // the engine's first stream waits for ev_in on the device; the call returns when the maps are complete
    if (sv_internal_process_after(g.engine, g.ev_in, in_l, in_r, g.W, g.d_disp, g.d_disp2) != SV_OK) {
        fprintf(stderr, "stereo_vision_hip: %s\n", sv_last_error(g.engine));
        return false;
    }
    sv::launch_dmap_and_cloud(g.d_disp, g.d_dmap, g.d_points, g.d_Q, g.W, g.H, g.stream);
    L_TRY(hipMemcpyAsync(g.points, g.d_points, N * sizeof(Double3), hipMemcpyDeviceToHost, g.stream));
    L_TRY(hipMemcpyAsync(g.h_dmap, g.d_dmap, N, hipMemcpyDeviceToHost, g.stream));
    if (resize) L_TRY(hipMemcpyAsync(g.colors.data(), g.d_bgra_l, N * 4, hipMemcpyDeviceToHost, g.stream));
    L_TRY(hipStreamSynchronize(g.stream));
    return true;
}

}  // namespace

extern "C" {

Double3 *generatePointCloud(unsigned char *left, unsigned char *right, char *CAMERA_CALIBRATION_YAML, int width, int height, bool kittiCalibration,
                            bool objectTracking, bool graphics, bool display, int scale, int pc_extrapolation, const char *YOLO_CFG,
                            const char *YOLO_WEIGHTS, const char *YOLO_CLASSES, bool removeSky, bool subsampling) {
    (void)kittiCalibration;
    (void)graphics;
    (void)display;
    (void)pc_extrapolation;  // the reference ignores the argument in favour of its global default 1 (stereo_vision.cpp:582)
    (void)YOLO_CFG;
    (void)YOLO_WEIGHTS;
    (void)YOLO_CLASSES;
    // Arguments 15 / 16 are never read: the reference's binding passes 14 (stereo_vision/sv.py:180,189), so their slots hold whatever
    // the caller left there.  removeSky has no effect in the reference anyway (dmapOLD.copyTo(dmapOLD, sky_mask) copies an image onto
    // itself, stereo_vision.cpp:604-607); subsampling comes from sv_legacy_set_subsampling.
    (void)removeSky;
    (void)subsampling;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!left || !right) return nullptr;
    if (!g.ready && !g.failed) {  // function-static init of the reference (stereo_vision.cpp:582): first call freezes the state
        if (width < 32 || height < 32 || scale < 1) {  // K1/K2 are divided by scale (:364-376); the reference would divide by zero
            fprintf(stderr, "stereo_vision_hip: bad width/height/scale (%d, %d, %d)\n", width, height, scale);
            return nullptr;
        }
        if (objectTracking) printf("\n** Object tracking requested: not provided by this library (detector weights are not part of the hot path)\n");
        else printf("\n** Object tracking disabled\n");
        g.ready = legacy_init(width, height, (float)scale, CAMERA_CALIBRATION_YAML, g_want_subsampling);
        g.failed = !g.ready;
    }
    if (!g.ready) return nullptr;
    // The reference wraps each call's width x height buffers and resizes them to the frozen out_img_size (:587-591).
    if (width < 1 || height < 1 || (size_t)width * height > ((size_t)1 << 26)) {
        fprintf(stderr, "stereo_vision_hip: bad frame size %dx%d\n", width, height);
        return nullptr;
    }
    if (!legacy_frame(left, right, width, height)) return nullptr;
    return g.points;
}

void clean(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.engine) sv_destroy(g.engine);
    if (g.ready || g.failed) (void)hipSetDevice(g.device);
    void *dptrs[] = {g.d_bgra_l, g.d_bgra_r, g.d_gray_l, g.d_gray_r, g.d_dmap, g.d_disp, g.d_disp2, g.d_points, g.d_Q, g.d_maps, g.d_rect_l, g.d_rect_r, g.d_src_l, g.d_src_r};
    for (void *p : dptrs)
        if (p) (void)hipFree(p);
    if (g.ev_in) (void)hipEventDestroy(g.ev_in);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    if (g.h_stage) (void)hipHostFree(g.h_stage);
    if (g.points) (void)hipHostFree(g.points);
    if (g.h_dmap) (void)hipHostFree(g.h_dmap);
    g = Legacy();
    printf("\n\nProgram exitted successfully!\n\n");  // stereo_vision.cpp:111 (the reference then calls exit(0); we return)
}

Uchar4 *getColor(void) { return g.colors.empty() ? nullptr : g.colors.data(); }

const unsigned char *sv_legacy_last_dmap(int *width, int *height) {
    if (width) *width = g.W;
    if (height) *height = g.H;
    return g.h_dmap;
}

const double *sv_legacy_Q(void) { return g.ready ? g.rect.Q : nullptr; }

void sv_legacy_set_rectify(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_want_rectify = on != 0;
}

void sv_legacy_set_subsampling(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_want_subsampling = on != 0;
}

void sv_legacy_set_device(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_want_device = device < 0 ? 0 : device;
}

const float *sv_legacy_rectify_maps(void) { return (g.ready && g.rectify) ? g.h_maps.data() : nullptr; }

int sv_legacy_last_gray(unsigned char *left, unsigned char *right) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready || !left || !right) return -1;
    const size_t N = (size_t)g.W * g.H;
    if (hipMemcpy(left, g.rectify ? g.d_rect_l : g.d_gray_l, N, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(right, g.rectify ? g.d_rect_r : g.d_gray_r, N, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}

int sv_reproject_batch_device(const float *disp, int batch, int width, int height, const double *Q16, const double *XR9, const double *XT3, unsigned char *dmap_out,
                              double *points_out) {
    if (!disp || !Q16 || !points_out || batch < 0 || width < 1 || height < 1) return SV_ERR_ARG;
    if (batch == 0) return SV_OK;
    if (sv::launch_reproject_batch(disp, batch, width, height, Q16, XR9, XT3, dmap_out, points_out, nullptr) != 0) return SV_ERR_HIP;
    return hipStreamSynchronize(nullptr) == hipSuccess ? SV_OK : SV_ERR_HIP;
}

int sv_disparity_to_u8_device(const float *disp, size_t count, unsigned char *dmap_out, void *stream) {
    if (!disp || !dmap_out) return SV_ERR_ARG;
    if (sv::launch_disp_to_u8(disp, count, dmap_out, static_cast<hipStream_t>(stream)) != 0) return SV_ERR_HIP;
    return SV_OK;
}

int sv_legacy_box_means(const int32_t *boxes, int n, double *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready || !g.points || !boxes || !out || n < 0) return -1;
    auto constrain = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int b = 0; b < n; b++) {  // stereo_vision.cpp:262-277
        const int x = boxes[4 * b], y = boxes[4 * b + 1], w = boxes[4 * b + 2], hgt = boxes[4 * b + 3];
        const int i_lb = constrain(x, 0, g.W - 1), i_ub = constrain(x + w, 0, g.W - 1);
        const int j_lb = constrain(y, 0, g.H - 1), j_ub = constrain(y + hgt, 0, g.H - 1);
        double X = 0, Y = 0, Z = 0;
        for (int i = i_lb; i < i_ub; i++)
            for (int j = j_lb; j < j_ub; ++j) {
                X += g.points[(size_t)j * g.W + i].x;
                Y += g.points[(size_t)j * g.W + i].y;
                Z += g.points[(size_t)j * g.W + i].z;
            }
        const double cnt = (double)((i_ub - i_lb) * (j_ub - j_lb));  // int product converted at the division, as in the reference
        out[3 * b] = X / cnt;
        out[3 * b + 1] = Y / cnt;
        out[3 * b + 2] = Z / cnt;
    }
    return 0;
}

} /* extern "C" */
