/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's ELAS hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libstereo_vision_hip.so) never links, loads or calls anything under oracle/.
 *
 * Parity pin: every stage of this restatement is compared bit for bit against the reference's own
 * serial LIBELAS compiled from /root/reference (oracle/_ref, see oracle/Makefile and
 * tests/test_oracle_vs_ref.py) and against the fixtures under tests/golden/ which that build produced
 * (tests/golden/make_golden.py).  The reference's own tests hold no golden vectors.
 */
#ifndef ORACLE_ELAS_ORACLE_H
#define ORACLE_ELAS_ORACLE_H

#include <stdint.h>

#include "elas_params.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Whole pipeline, Elas::process semantics (elas.cpp:31-150).  Returns seconds per call. */
double orc_process(const elas_params *p, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride, float *D1, float *D2, int canonical, int reps);

/* Whole pipeline keeping every intermediate; read them back with orc_size/orc_get.  Stage names:
 *   desc1 desc2 (u8 H*W*16) dcan_raw (i16 Hc*Wc) dcan_dims (i32 2) support (i32 n*3)
 *   tri1 tri2 (i32 nt*3) planes1 planes2 (f32 nt*6) grid1 grid2 (i32 gh*gw*(disp_max+2)) grid_dims (i32 3)
 *   wta{1,2} lr{1,2} speckle{1,2} gap{1,2} amean{1,2} final{1,2} (f32 H*W)
 * Returns the number of support points (<0 on error). */
int orc_run_stages(const elas_params *p, const uint8_t *I1, const uint8_t *I2, int W, int H, int stride);
long orc_size(const char *name);
long orc_get(const char *name, void *out, long cap);

/* Individual stages (same arithmetic as the pipeline above). */
void orc_descriptor(const uint8_t *I, int W, int H, int stride, uint8_t *desc /* H*W*16, fully written */);
void orc_support_raw(const elas_params *p, const uint8_t *desc1, const uint8_t *desc2, int W, int H, int16_t *dcan /* Hc*Wc */);
int orc_support_filter(const elas_params *p, int16_t *dcan /* in/out */, int W, int H, int32_t *support /* cap*3 */, int cap);
int orc_delaunay(const float *xy, int n, int32_t *tri_out, int cap);
void orc_planes(const int32_t *support, const int32_t *tri, int nt, float *planes /* nt*6: t1a t1b t1c t2a t2b t2c */);
void orc_grid(const elas_params *p, const int32_t *support, int n, int W, int H, int right_image, int32_t *grid);
void orc_dense(const elas_params *p, const int32_t *support, const int32_t *tri, const float *planes, int nt, const int32_t *grid,
               const uint8_t *desc1, const uint8_t *desc2, int W, int H, int right_image, float *D);
void orc_lr_check(const elas_params *p, float *D1, float *D2, int W, int H);
void orc_speckle(const elas_params *p, float *D, int W, int H);
void orc_gap(const elas_params *p, float *D, int W, int H);
void orc_adaptive_mean(float *D, int W, int H);
void orc_median(float *D, int W, int H);

#ifdef __cplusplus
}
#endif
#endif
