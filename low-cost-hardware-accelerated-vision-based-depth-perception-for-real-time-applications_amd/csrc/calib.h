// Calibration front-end of the legacy entry point, without OpenCV:
//   * reader for the OpenCV-FileStorage YAML subset the reference's calibration files use
//     (reference: data/calibration/kitti_2011_09_26.yml, read at src/serial_includes/main/stereo_vision.cpp:528-537)
//   * disparity-to-depth matrix Q as cv::stereoRectify(K1,D1,K2,D2,size,R,T,...,CALIB_ZERO_DISPARITY, alpha=0, size)
//     computes it (call site: stereo_vision.cpp:439).  OpenCV is an un-vendored third-party dependency of the
//     reference (CI pins 4.4.0); what follows restates the published algorithm (Bouguet's rectification as implemented by
//     cvStereoRectify): parity unpinned except for the Q printed in the reference's own comment (stereo_vision.cpp:211-215).
#pragma once

#include <string>

namespace sv {

struct Calibration {
    double K1[9], K2[9];
    double D1[5], D2[5];
    double R[9];
    double T[3];
    double XR[9], XT[3];
    bool has_xr = false, has_xt = false;
};

// Returns false (and sets err) if the file cannot be read or a required entry (K1 K2 D1 D2 R T) is missing.
bool load_calibration_yaml(const char *path, Calibration &c, std::string &err);

struct Rectification {
    double R1[9], R2[9];
    double P1[12], P2[12];
    double Q[16];
};

// image_size = size the calibration refers to, new_size = size of the rectified images (the driver passes the same
// value for both, stereo_vision.cpp:524-525,547), alpha as cv::stereoRectify (the driver passes 0).
void stereo_rectify(const Calibration &c, int image_w, int image_h, int new_w, int new_h, double alpha, Rectification &out);

// cv::initUndistortRectifyMap(K, D, R, P, (w, h), CV_32F, mapx, mapy) (stereo_vision.cpp:477-478); maps are [h][w] floats.
bool init_undistort_rectify_map(const double *K, const double *D, const double *R, const double *P, int w, int h, float *mapx, float *mapy);

}  // namespace sv
