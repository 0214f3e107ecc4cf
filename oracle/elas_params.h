/*
 * TEST INFRASTRUCTURE (oracle side).  Plain-C parameter block shared by the CPU
 * restatement (elas_oracle.cpp) and the reference harness (ref_harness.cpp).
 *
 * Field-for-field mirror of `Elas::parameters`
 *   (reference: src/serial_includes/elas/elas.h:60-145)
 * with every bool widened to int32 so that the struct is a flat array of 23
 * 4-byte words and can be described by one ctypes.Structure on the Python side.
 * The product's `sv_params` (include/stereo_vision_hip.h) has the identical layout.
 */
#ifndef ORACLE_ELAS_PARAMS_H
#define ORACLE_ELAS_PARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct elas_params {
    int32_t disp_min;              /* elas.h:61  */
    int32_t disp_max;              /* elas.h:62  (D = disp_max + 1) */
    float support_threshold;       /* elas.h:63  */
    int32_t support_texture;       /* elas.h:64  */
    int32_t candidate_stepsize;    /* elas.h:65  */
    int32_t incon_window_size;     /* elas.h:66  */
    int32_t incon_threshold;       /* elas.h:67  */
    int32_t incon_min_support;     /* elas.h:68  */
    int32_t add_corners;           /* elas.h:69  (bool) */
    int32_t grid_size;             /* elas.h:70  */
    float beta;                    /* elas.h:71  */
    float gamma;                   /* elas.h:72  */
    float sigma;                   /* elas.h:73  */
    float sradius;                 /* elas.h:74  */
    int32_t match_texture;         /* elas.h:75  */
    int32_t lr_threshold;          /* elas.h:76  */
    float speckle_sim_threshold;   /* elas.h:77  */
    int32_t speckle_size;          /* elas.h:78  */
    int32_t ipol_gap_width;        /* elas.h:79  */
    int32_t filter_median;         /* elas.h:80  (bool) */
    int32_t filter_adaptive_mean;  /* elas.h:81  (bool) */
    int32_t postprocess_only_left; /* elas.h:82  (bool) */
    int32_t subsampling;           /* elas.h:83  (bool) */
} elas_params;

enum { ELAS_ROBOTICS = 0, ELAS_MIDDLEBURY = 1 };

/* Presets: elas.h:92-115 (ROBOTICS) and elas.h:119-143 (MIDDLEBURY). */
static inline void elas_params_preset(elas_params *p, int setting) {
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
    if (setting == ELAS_ROBOTICS) {
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
    }
}

/* The configuration the reference driver runs (stereo_vision.cpp:307-311):
 * MIDDLEBURY + postprocess_only_left + filter_adaptive_mean (median stays on). */
static inline void elas_params_driver(elas_params *p, int disp_max) {
    elas_params_preset(p, ELAS_MIDDLEBURY);
    p->postprocess_only_left = 1;
    p->filter_adaptive_mean = 1;
    p->disp_max = disp_max;
}

#ifdef __cplusplus
}
#endif
#endif
