// See calib.h.  Double precision throughout, except where OpenCV itself stores intermediate points as float32
// (the corner / 9x9 grid points of cvStereoRectify and icvGetRectangles), which is reproduced.
#include "calib.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <sstream>
#include <vector>

namespace sv {

// ------------------------------------------------------------------------------------------------------------
// YAML subset:   NAME: !!opencv-matrix\n rows: r\n cols: c\n dt: d\n data: [ a, b, ... ]      and      NAME: [ a, b, c ]
// ------------------------------------------------------------------------------------------------------------
static bool read_numbers_after(const std::string &text, size_t pos, std::vector<double> &out) {
    size_t lb = text.find('[', pos);
    if (lb == std::string::npos) return false;
    size_t rb = text.find(']', lb);
    if (rb == std::string::npos) return false;
    std::string body = text.substr(lb + 1, rb - lb - 1);
    for (char &ch : body)
        if (ch == ',' || ch == '\n' || ch == '\r' || ch == '\t') ch = ' ';
    std::istringstream ss(body);
    std::string tok;
    out.clear();
    while (ss >> tok) out.push_back(strtod(tok.c_str(), nullptr));
    return true;
}

static bool find_entry(const std::string &text, const char *name, std::vector<double> &vals) {
    const std::string key = std::string(name) + ":";
    size_t pos = 0;
    while ((pos = text.find(key, pos)) != std::string::npos) {
        const bool at_line_start = pos == 0 || text[pos - 1] == '\n';
        if (at_line_start) {
            // the value list is the first [...] after the key and before the next top-level key
            size_t next_key = text.find("\n", pos);
            size_t lb = text.find('[', pos);
            (void)next_key;
            if (lb == std::string::npos) return false;
            return read_numbers_after(text, pos, vals);
        }
        pos += key.size();
    }
    return false;
}

bool load_calibration_yaml(const char *path, Calibration &c, std::string &err) {
    std::ifstream f(path ? path : "");
    if (!f) {
        err = std::string("cannot open calibration file: ") + (path ? path : "(null)");
        return false;
    }
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string text = buf.str();
    struct Want {
        const char *name;
        double *dst;
        int n;
        bool required;
        bool *flag;
    } wants[] = {{"K1", c.K1, 9, true, nullptr}, {"K2", c.K2, 9, true, nullptr}, {"D1", c.D1, 5, true, nullptr}, {"D2", c.D2, 5, true, nullptr},
                 {"R", c.R, 9, true, nullptr},   {"T", c.T, 3, true, nullptr},   {"XR", c.XR, 9, false, &c.has_xr}, {"XT", c.XT, 3, false, &c.has_xt}};
    for (const Want &w : wants) {
        std::vector<double> v;
        const bool ok = find_entry(text, w.name, v);
        if (!ok || (int)v.size() < (w.n == 5 ? 4 : w.n)) {
            if (w.required) {
                err = std::string("calibration entry missing or malformed: ") + w.name;
                return false;
            }
            continue;
        }
        for (int i = 0; i < w.n; i++) w.dst[i] = i < (int)v.size() ? v[i] : 0.0;
        if (w.flag) *w.flag = true;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// small 3x3 helpers
// ------------------------------------------------------------------------------------------------------------
static void mat3_mul(const double *A, const double *B, double *C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof(t));
}

static void mat3_mul_bt(const double *A, const double *B, double *C) {  // A * B^T
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
    memcpy(C, t, sizeof(t));
}

static void mat3_vec(const double *A, const double *v, double *o) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
    memcpy(o, t, sizeof(t));
}

static bool mat3_inv_t(const double *A, double *o) {  // o = (A^-1)^T = cofactor matrix / det
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (fabs(det) < 1e-300) return false;
    const double id = 1.0 / det;
    o[0] = c00 * id;
    o[1] = c01 * id;
    o[2] = c02 * id;
    o[3] = (A[2] * A[7] - A[1] * A[8]) * id;
    o[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    o[5] = (A[1] * A[6] - A[0] * A[7]) * id;
    o[6] = (A[1] * A[5] - A[2] * A[4]) * id;
    o[7] = (A[2] * A[3] - A[0] * A[5]) * id;
    o[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

// nearest rotation (orthogonal polar factor U*V^T of the SVD that cvRodrigues2 applies to a matrix argument),
// by Newton iteration X <- (X + X^-T)/2
static void orthonormalize(const double *Rin, double *Rout) {
    double X[9];
    memcpy(X, Rin, sizeof(X));
    for (int it = 0; it < 50; it++) {
        double Y[9];
        if (!mat3_inv_t(X, Y)) break;
        double diff = 0;
        for (int i = 0; i < 9; i++) {
            const double n = 0.5 * (X[i] + Y[i]);
            diff = std::max(diff, fabs(n - X[i]));
            X[i] = n;
        }
        if (diff < 1e-16) break;
    }
    memcpy(Rout, X, sizeof(X));
}

static void rodrigues_vec_to_mat(const double *r, double *R) {
    const double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(R, I, sizeof(I));
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rx[k];
}

static void rodrigues_mat_to_vec(const double *Rin, double *r) {
    double R[9];
    orthonormalize(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    const double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            const double n = theta / sqrt(rx * rx + ry * ry + rz * rz);
            r[0] = rx * n;
            r[1] = ry * n;
            r[2] = rz * n;
        }
    } else {
        const double vth = 1 / (2 * s) * theta;
        r[0] = rx * vth;
        r[1] = ry * vth;
        r[2] = rz * vth;
    }
}

// cv::undistortPoints for the 5-coefficient model (k1 k2 p1 p2 k3), 5 fixed-point iterations, optional rectification
// rotation R (3x3) and new camera matrix P (fx, fy, cx, cy of its left 3x3 block); output rounded to float32 like OpenCV's
// CV_32FC2 destination.
static void undistort_point(const double *K, const double *D, const double *R, const double *P, float xin, float yin, float &xo, float &yo) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double x = ((double)xin - cx) / fx, y = ((double)yin - cy) / fy;
    const double x0 = x, y0 = y;
    if (D) {
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2);
            if (icdist < 0) {
                x = x0;
                y = y0;
                break;
            }
            const double deltaX = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
            const double deltaY = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
    }
    double RR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) memcpy(RR, R, sizeof(RR));
    if (P) {
        const double PP[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
        mat3_mul(PP, RR, RR);
    }
    const double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5];
    const double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
    xo = (float)(xx * ww);
    yo = (float)(yy * ww);
}

struct RectF {
    float x, y, w, h;
};

static void get_rectangles(const double *K, const double *D, const double *R, const double *P, int iw, int ih, RectF &inner, RectF &outer) {
    const int N = 9;
    float iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            float px, py;
            undistort_point(K, D, R, P, (float)x * iw / (N - 1), (float)y * ih / (N - 1), px, py);
            oX0 = std::min(oX0, px);
            oX1 = std::max(oX1, px);
            oY0 = std::min(oY0, py);
            oY1 = std::max(oY1, py);
            if (x == 0) iX0 = std::max(iX0, px);
            if (x == N - 1) iX1 = std::min(iX1, px);
            if (y == 0) iY0 = std::max(iY0, py);
            if (y == N - 1) iY1 = std::min(iY1, py);
        }
    inner = RectF{iX0, iY0, iX1 - iX0, iY1 - iY0};
    outer = RectF{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

int g_rectify_variant = 1;  // 1: OpenCV >= 3.4.2 / 4.x focal-length rule (mean focal * size ratio, corners at nx, ny); 0: older rule

void stereo_rectify(const Calibration &c, int image_w, int image_h, int new_w, int new_h, double alpha, Rectification &out) {
    double om[3], r_r[9], t[3], uu[3] = {0, 0, 0}, ww[3], wR[9], Ri[9];
    rodrigues_mat_to_vec(c.R, om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5;  // average rotation
    rodrigues_vec_to_mat(om, r_r);
    mat3_vec(r_r, c.T, t);
    const int idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    const double cc = t[idx], nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = cc > 0 ? 1 : -1;
    ww[0] = t[1] * uu[2] - t[2] * uu[1];
    ww[1] = t[2] * uu[0] - t[0] * uu[2];
    ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) {
        const double sc = acos(fabs(cc) / nt) / nw;
        for (int i = 0; i < 3; i++) ww[i] *= sc;
    }
    rodrigues_vec_to_mat(ww, wR);
    mat3_mul_bt(wR, r_r, Ri);
    memcpy(out.R1, Ri, sizeof(Ri));
    mat3_mul(wR, r_r, Ri);
    memcpy(out.R2, Ri, sizeof(Ri));
    mat3_vec(Ri, c.T, t);

    const double nx = image_w, ny = image_h;
    if (new_w * new_h == 0) {
        new_w = image_w;
        new_h = image_h;
    }
    double fc_new;
    const double edge_x = g_rectify_variant ? nx : nx - 1, edge_y = g_rectify_variant ? ny : ny - 1;
    if (g_rectify_variant) {
        const double ratio_x = (double)new_w / image_w / 2, ratio_y = (double)new_h / image_h / 2;
        const double ratio = idx == 1 ? ratio_x : ratio_y;
        const int k = (idx ^ 1) * 3 + (idx ^ 1);
        fc_new = (c.K1[k] + c.K2[k]) * ratio;
    } else {
        fc_new = DBL_MAX;
        for (int k = 0; k < 2; k++) {
            const double *A = k == 0 ? c.K1 : c.K2, *Dk = k == 0 ? c.D1 : c.D2;
            double fc = A[(idx ^ 1) * 3 + (idx ^ 1)];
            if (Dk[0] < 0) fc *= 1 + Dk[0] * (nx * nx + ny * ny) / (4 * fc * fc);
            fc_new = std::min(fc_new, fc);
        }
    }
    double cc_new[2][2];
    for (int k = 0; k < 2; k++) {
        const double *A = k == 0 ? c.K1 : c.K2, *Dk = k == 0 ? c.D1 : c.D2;
        const double *Rk = k == 0 ? out.R1 : out.R2;
        double sx = 0, sy = 0;
        for (int i = 0; i < 4; i++) {
            const int j = i < 2 ? 0 : 1;
            float px, py;
            undistort_point(A, Dk, nullptr, nullptr, (float)((i % 2) * edge_x), (float)(j * edge_y), px, py);
            // project (px, py, 1) with rotation Rk, focal fc_new, principal point 0 (cvProjectPoints2), float32 destination
            const double X = Rk[0] * px + Rk[1] * py + Rk[2], Y = Rk[3] * px + Rk[4] * py + Rk[5], Z = Rk[6] * px + Rk[7] * py + Rk[8];
            const double z = Z ? 1. / Z : 1;
            sx += (double)(float)(fc_new * X * z);
            sy += (double)(float)(fc_new * Y * z);
        }
        cc_new[k][0] = edge_x / 2 - sx / 4;
        cc_new[k][1] = edge_y / 2 - sy / 4;
    }
    // CALIB_ZERO_DISPARITY (stereo_vision.cpp:439)
    cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5;
    cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5;

    double pp[12] = {0};
    pp[0] = pp[5] = fc_new;
    pp[2] = cc_new[0][0];
    pp[6] = cc_new[0][1];
    pp[10] = 1;
    memcpy(out.P1, pp, sizeof(pp));
    pp[2] = cc_new[1][0];
    pp[6] = cc_new[1][1];
    pp[idx * 4 + 3] = t[idx] * fc_new;
    memcpy(out.P2, pp, sizeof(pp));

    alpha = std::min(alpha, 1.);
    RectF inner1, inner2, outer1, outer2;
    get_rectangles(c.K1, c.D1, out.R1, out.P1, image_w, image_h, inner1, outer1);
    get_rectangles(c.K2, c.D2, out.R2, out.P2, image_w, image_h, inner2, outer2);
    {
        const double cx1_0 = cc_new[0][0], cy1_0 = cc_new[0][1], cx2_0 = cc_new[1][0], cy2_0 = cc_new[1][1];
        const double cx1 = new_w * cx1_0 / image_w, cy1 = new_h * cy1_0 / image_h;
        const double cx2 = new_w * cx2_0 / image_w, cy2 = new_h * cy2_0 / image_h;
        double s = 1.;
        if (alpha >= 0) {
            double s0 = std::max(std::max(std::max(cx1 / (cx1_0 - inner1.x), cy1 / (cy1_0 - inner1.y)), (new_w - cx1) / (inner1.x + inner1.w - cx1_0)),
                                 (new_h - cy1) / (inner1.y + inner1.h - cy1_0));
            s0 = std::max(std::max(std::max(std::max(cx2 / (cx2_0 - inner2.x), cy2 / (cy2_0 - inner2.y)), (new_w - cx2) / (inner2.x + inner2.w - cx2_0)),
                                   (new_h - cy2) / (inner2.y + inner2.h - cy2_0)),
                          s0);
            double s1 = std::min(std::min(std::min(cx1 / (cx1_0 - outer1.x), cy1 / (cy1_0 - outer1.y)), (new_w - cx1) / (outer1.x + outer1.w - cx1_0)),
                                 (new_h - cy1) / (outer1.y + outer1.h - cy1_0));
            s1 = std::min(std::min(std::min(std::min(cx2 / (cx2_0 - outer2.x), cy2 / (cy2_0 - outer2.y)), (new_w - cx2) / (outer2.x + outer2.w - cx2_0)),
                                   (new_h - cy2) / (outer2.y + outer2.h - cy2_0)),
                          s1);
            s = s0 * (1 - alpha) + s1 * alpha;
        }
        fc_new *= s;
        cc_new[0][0] = cx1;
        cc_new[0][1] = cy1;
        cc_new[1][0] = cx2;
        cc_new[1][1] = cy2;
        out.P1[0] = out.P1[5] = fc_new;
        out.P1[2] = cx1;
        out.P1[6] = cy1;
        out.P2[0] = out.P2[5] = fc_new;
        out.P2[2] = cx2;
        out.P2[6] = cy2;
        out.P2[idx * 4 + 3] = s * out.P2[idx * 4 + 3];
    }
    const double q[16] = {1, 0, 0, -cc_new[0][0], 0, 1, 0, -cc_new[0][1], 0, 0, 0, fc_new, 0, 0, -1. / t[idx],
                          (idx == 0 ? cc_new[0][0] - cc_new[1][0] : cc_new[0][1] - cc_new[1][1]) / t[idx]};
    memcpy(out.Q, q, sizeof(q));
}

}  // namespace sv

// test hook (tests/test_calib.py): returns Q for a calibration file
namespace sv {

static bool mat3_inverse(const double *A, double *I) {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    I[0] = c00 * id;
    I[1] = (A[2] * A[7] - A[1] * A[8]) * id;
    I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = c01 * id;
    I[4] = (A[0] * A[8] - A[2] * A[6]) * id;
    I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = c02 * id;
    I[7] = (A[1] * A[6] - A[0] * A[7]) * id;
    I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

// cv::initUndistortRectifyMap(K, D, R, P, size, CV_32F, mapx, mapy) (call sites: stereo_vision.cpp:477-478): for every pixel
// (u, v) of the rectified image the position in the distorted source image.  Restates the documented algorithm (pinhole +
// k1,k2,p1,p2,k3 model): [x y w]^T = (P[:, :3] * R)^-1 * [u v 1]^T, x /= w, y /= w, radial + tangential distortion, back through K.
// Double arithmetic, row-wise increments like OpenCV's scalar loop.  OpenCV is not available here: parity unpinned.
bool init_undistort_rectify_map(const double *K, const double *D, const double *R, const double *P, int w, int h, float *mapx, float *mapy) {
    double Ar[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]}, ArR[9], ir[9];
    mat3_mul(Ar, R, ArR);
    if (!mat3_inverse(ArR, ir)) return false;
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
            const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            mapx[(size_t)i * w + j] = (float)(fx * xd + u0);
            mapy[(size_t)i * w + j] = (float)(fy * yd + v0);
        }
    }
    return true;
}

}  // namespace sv

extern "C" int sv_debug_stereo_rectify(const char *yaml, int image_w, int image_h, double scale, int variant, double *Q16, double *P1P2_24) {
    sv::Calibration c;
    std::string err;
    if (!sv::load_calibration_yaml(yaml, c, err)) return -1;
    for (int i = 0; i < 6; i++) {  // first two rows of K1, K2 (stereo_vision.cpp:364-376)
        c.K1[i] /= scale;
        c.K2[i] /= scale;
    }
    sv::g_rectify_variant = variant;
    sv::Rectification r;
    sv::stereo_rectify(c, image_w, image_h, image_w, image_h, 0.0, r);
    sv::g_rectify_variant = 1;
    memcpy(Q16, r.Q, sizeof(r.Q));
    if (P1P2_24) {
        memcpy(P1P2_24, r.P1, sizeof(r.P1));
        memcpy(P1P2_24 + 12, r.P2, sizeof(r.P2));
    }
    return 0;
}

extern "C" int sv_debug_undistort_map(const double *K9, const double *D5, const double *R9, const double *P12, int w, int h, float *mapx, float *mapy) {
    if (!K9 || !D5 || !R9 || !P12 || !mapx || !mapy || w < 1 || h < 1) return -1;
    return sv::init_undistort_rectify_map(K9, D5, R9, P12, w, h, mapx, mapy) ? 0 : -1;
}
