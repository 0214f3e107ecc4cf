// Front-/back-end maps of the legacy entry point, on the GPU:
//   BGRA -> gray        cv::cvtColor(COLOR_BGRA2GRAY) for 8-bit images, OpenCV 4.x fixed point (15-bit weights
//                       B 3735, G 19235, R 9798, rounding 1<<14)              call site: stereo_vision.cpp:338-339
//   resize              cv::resize(INTER_LINEAR) of a frame whose size differs from the frozen one           :590-591
//   f32 -> u8 x4        leftdpf.convertTo(dmap, CV_8UC1, 4.0) = saturate(round-half-even(4*d))       :316
//   reprojection        pos = Q*[i j d 1]^T, (X,Y,Z) = pos.xyz / pos.w in double                     :233-256
// OpenCV is an un-vendored dependency of the reference: these restate its documented arithmetic (parity unpinned).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sv {

__global__ __launch_bounds__(256) void k_bgra_to_gray(const uchar4 *__restrict__ l, const uchar4 *__restrict__ r, uint8_t *__restrict__ gl, uint8_t *__restrict__ gr, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uchar4 a = l[i], b = r[i];  // x = B, y = G, z = R, w = A
    gl[i] = (uint8_t)((a.x * 3735 + a.y * 19235 + a.z * 9798 + 16384) >> 15);
    gr[i] = (uint8_t)((b.x * 3735 + b.y * 19235 + b.z * 9798 + 16384) >> 15);
}

void launch_bgra_to_gray(const unsigned char *bgra_l, const unsigned char *bgra_r, unsigned char *gray_l, unsigned char *gray_r, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_bgra_to_gray, dim3((n + 255) / 256), dim3(256), 0, st, (const uchar4 *)bgra_l, (const uchar4 *)bgra_r, gray_l, gray_r, n);
}

// cv::resize(src, dst, dsize) with the default INTER_LINEAR for 8UC4 images (stereo_vision.cpp:590-591 - the frame a later call
// hands over is resized to the size the first call froze), OpenCV 4.x's generic path restated: per destination column
// fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx (clamped to the first / last source column with fx = 0),
// 11-bit coefficients a = cvRound((1 - fx) * 2048), cvRound(fx * 2048); rows likewise, their indices clipped instead;
// horizontal pass in int (S0*a0 + S1*a1), vertical pass ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2.  Exact 2x
// decimation in both directions takes OpenCV's INTER_AREA shortcut ((sum of the 2x2 block + 2) >> 2).  Equal sizes are a
// plain copy and never get here.  Parity unpinned, like the other OpenCV restatements of this file.
__global__ __launch_bounds__(256) void k_resize_bgra(const uchar4 *__restrict__ src, int sw, int sh, uchar4 *__restrict__ dst, int dw, int dh, double scale_x, double scale_y,
                                                     int area2) {
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    if (area2) {
        const uchar4 a = src[(size_t)(2 * dy) * sw + 2 * dx], b = src[(size_t)(2 * dy) * sw + 2 * dx + 1];
        const uchar4 c = src[(size_t)(2 * dy + 1) * sw + 2 * dx], e = src[(size_t)(2 * dy + 1) * sw + 2 * dx + 1];
        dst[(size_t)dy * dw + dx] = make_uchar4((unsigned char)((a.x + b.x + c.x + e.x + 2) >> 2), (unsigned char)((a.y + b.y + c.y + e.y + 2) >> 2),
                                                (unsigned char)((a.z + b.z + c.z + e.z + 2) >> 2), (unsigned char)((a.w + b.w + c.w + e.w + 2) >> 2));
        return;
    }
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) fx = 0.f, sx = 0;
    if (sx >= sw - 1) fx = 0.f, sx = sw - 1;
    const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
    const int sx1 = min(sx + 1, sw - 1);
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
    const int y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
    const uchar4 p00 = src[(size_t)y0 * sw + sx], p01 = src[(size_t)y0 * sw + sx1], p10 = src[(size_t)y1 * sw + sx], p11 = src[(size_t)y1 * sw + sx1];
    auto mix = [&](int s00, int s01, int s10, int s11) -> unsigned char {
        const int h0 = s00 * a0 + s01 * a1, h1 = s10 * a0 + s11 * a1;
        return (unsigned char)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    };
    dst[(size_t)dy * dw + dx] = make_uchar4(mix(p00.x, p01.x, p10.x, p11.x), mix(p00.y, p01.y, p10.y, p11.y), mix(p00.z, p01.z, p10.z, p11.z), mix(p00.w, p01.w, p10.w, p11.w));
}

void launch_resize_bgra(const unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh, hipStream_t st) {
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);  // resize.cpp: scale = 1 / inv_scale
    const int area2 = (sw == 2 * dw && sh == 2 * dh) ? 1 : 0;
    hipLaunchKernelGGL(k_resize_bgra, dim3((dw + 255) / 256, dh), dim3(256), 0, st, (const uchar4 *)src, sw, sh, (uchar4 *)dst, dw, dh, scale_x, scale_y, area2);
}

// cv::remap(src, dst, mapx, mapy, INTER_LINEAR) for 8-bit single-channel images and CV_32FC1 maps, BORDER_CONSTANT(0)
// (the call the reference has commented out at stereo_vision.cpp:341): OpenCV's fixed-point bilinear - the map converted to 5
// fractional bits with round-half-even, weights (32-fx)(32-fy)*32 ... (they sum to 2^15 exactly), result (sum + 2^14) >> 15.
__global__ __launch_bounds__(256) void k_remap_gray(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const float *__restrict__ mapx, const float *__restrict__ mapy,
                                                    int W, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= W) return;
    const size_t p = (size_t)j * W + i;
    const int sx = __float2int_rn(mapx[p] * 32.0f), sy = __float2int_rn(mapy[p] * 32.0f);  // cvRound
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    auto at = [&](int x, int y) -> int { return (x >= 0 && x < W && y >= 0 && y < H) ? (int)src[(size_t)y * W + x] : 0; };
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const int v = at(ix, iy) * w00 + at(ix + 1, iy) * w01 + at(ix, iy + 1) * w10 + at(ix + 1, iy + 1) * w11;
    dst[p] = (uint8_t)((v + (1 << 14)) >> 15);
}

void launch_remap_gray(const unsigned char *src, unsigned char *dst, const float *mapx, const float *mapy, int W, int H, hipStream_t st) {
    hipLaunchKernelGGL(k_remap_gray, dim3((W + 255) / 256, H), dim3(256), 0, st, src, dst, mapx, mapy, W, H);
}

__global__ __launch_bounds__(256) void k_dmap_cloud(const float *__restrict__ disp, uint8_t *__restrict__ dmap, double *__restrict__ pts, const double *__restrict__ Q, int W, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= W) return;
    const size_t p = (size_t)j * W + i;
    int v = __float2int_rn(disp[p] * 4.0f);  // cvRound: round half to even
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dmap[p] = (uint8_t)v;
    const double x = (double)i, y = (double)j, d = (double)v;
    double pos[4];
#pragma unroll
    for (int r = 0; r < 4; r++) pos[r] = ((Q[4 * r] * x + Q[4 * r + 1] * y) + Q[4 * r + 2] * d) + Q[4 * r + 3];
    pts[3 * p] = pos[0] / pos[3];
    pts[3 * p + 1] = pos[1] / pos[3];
    pts[3 * p + 2] = pos[2] / pos[3];
}

void launch_dmap_and_cloud(const float *disp, unsigned char *dmap, double *points, const double *Q16, int W, int H, hipStream_t st) {
    hipLaunchKernelGGL(k_dmap_cloud, dim3((W + 255) / 256, H), dim3(256), 0, st, disp, dmap, points, Q16, W, H);
}

// leftdpf.convertTo(dmap, CV_8UC1, 4.0) alone (stereo_vision.cpp:316), four pixels per thread: what a consumer that wants the driver's
// 8-bit disparity image - e.g. a gather of finished maps over xGMI, a quarter of the float maps' bytes - takes instead of D1.
__global__ __launch_bounds__(256) void k_disp_to_u8(const float *__restrict__ disp, uint8_t *__restrict__ out, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    auto cv8 = [](float d) -> uint32_t {
        int v = __float2int_rn(d * 4.0f);  // cvRound: round half to even; saturate_cast<uchar>
        return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    };
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(disp + i) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out + i) & 3) == 0)) {
        const float4 d = *reinterpret_cast<const float4 *>(disp + i);
        *reinterpret_cast<uint32_t *>(out + i) = cv8(d.x) | (cv8(d.y) << 8) | (cv8(d.z) << 16) | (cv8(d.w) << 24);
    } else {
        for (size_t j = i; j < n && j < i + 4; j++) out[j] = (uint8_t)cv8(disp[j]);
    }
}

int launch_disp_to_u8(const float *disp, size_t n, unsigned char *out, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_disp_to_u8, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, disp, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Batched form with the CUDA variant's optional robot-frame transform (parallel_includes/main/stereo_vision.cu:188-212:
// point = XR * (X, Y, Z) + XT); Q / XR / XT travel as kernel arguments.
struct ReprojectArgs {
    double Q[16], XR[9], XT[3];
    int has_xf;
};

__global__ __launch_bounds__(256) void k_reproject_batch(const float *__restrict__ disp, uint8_t *__restrict__ dmap, double *__restrict__ pts, ReprojectArgs a, int W, int H) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= W) return;
    const size_t p = ((size_t)blockIdx.z * H + j) * W + i;
    int v = __float2int_rn(disp[p] * 4.0f);  // convertTo(CV_8UC1, 4.0): round half to even, saturate
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (dmap) dmap[p] = (uint8_t)v;
    const double x = (double)i, y = (double)j, d = (double)v;
    double pos[4];
#pragma unroll
    for (int r = 0; r < 4; r++) pos[r] = ((a.Q[4 * r] * x + a.Q[4 * r + 1] * y) + a.Q[4 * r + 2] * d) + a.Q[4 * r + 3];
    double X = pos[0] / pos[3], Y = pos[1] / pos[3], Z = pos[2] / pos[3];
    if (a.has_xf) {
        const double px = ((a.XR[0] * X + a.XR[1] * Y) + a.XR[2] * Z) + a.XT[0];
        const double py = ((a.XR[3] * X + a.XR[4] * Y) + a.XR[5] * Z) + a.XT[1];
        const double pz = ((a.XR[6] * X + a.XR[7] * Y) + a.XR[8] * Z) + a.XT[2];
        X = px, Y = py, Z = pz;
    }
    pts[3 * p] = X;
    pts[3 * p + 1] = Y;
    pts[3 * p + 2] = Z;
}

int launch_reproject_batch(const float *disp, int batch, int W, int H, const double *Q16, const double *XR9, const double *XT3, unsigned char *dmap, double *points,
                           hipStream_t st) {
    ReprojectArgs a;
    for (int i = 0; i < 16; i++) a.Q[i] = Q16[i];
    a.has_xf = (XR9 || XT3) ? 1 : 0;
    for (int i = 0; i < 9; i++) a.XR[i] = XR9 ? XR9[i] : (i % 4 == 0 ? 1.0 : 0.0);
    for (int i = 0; i < 3; i++) a.XT[i] = XT3 ? XT3[i] : 0.0;
    hipLaunchKernelGGL(k_reproject_batch, dim3((W + 255) / 256, H, batch), dim3(256), 0, st, disp, dmap, points, a, W, H);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace sv
