#!/usr/bin/env python3
"""profiles/pmc_traffic.json (HBM bytes per pair and kernel, read by bench.py for roofline.traffic) from a table written by
tools/summarize_rocprof.py with the FETCH_SIZE and WRITE_SIZE passes merged in:

  python tools/pmc_traffic.py profiles/r01h_serial_kernel_stats_pmc.csv --pairs-per-launch 32 --note "tools/profile_run.py --chunk 32 --batch 64"
"""
import argparse
import csv
import json
import os

NAMES = {"k_filter_classify": "support_filter", "k_filter_resolve": "support_filter", "k_filter_vertical": "support_filter", "k_filter_horizontal": "support_filter", "k_filter_collect": "support_filter", "k_filter_count": "support_filter", "k_filter_corners": "support_filter", "k_dense": "dense_match", "k_support": "support_match", "k_descriptor": "descriptor", "k_sobel": "descriptor", "k_amean": "adaptive_mean",
         "k_amean_sub": "adaptive_mean", "k_raster_tiles": "triangles_raster", "k_lr": "lr_check", "k_lr2": "lr_check", "k_median": "median", "k_planes": "plane_fit",
         "k_ccl_band": "ccl_band", "k_gap_cols": "gap_cols", "k_gap_rows": "gap_rows", "k_grid_mark": "grid_mark", "k_grid_dilate": "grid_dilate",
         "k_raster": "triangles_raster_fallback", "k_ccl_border": "ccl_finish", "k_ccl_total": "ccl_finish", "k_ccl_apply": "ccl_finish", "k_ccl_slow": "ccl_finish",
         "dg::k_delaunay_blob": "delaunay_gpu", "dg::k_delaunay_resident": "delaunay_gpu", "dg::k_dg_prepare_large_blob": "delaunay_gpu", "dg::k_dgl_subtrees_blob": "delaunay_gpu", "dg::k_dgl_top_blob": "delaunay_gpu"}
WIDE_READERS = {"k_descriptor"}  # round 1: k_dense, k_support, k_descriptor read / re-read 16-byte descriptors (dwordx4 per lane)
ap = argparse.ArgumentParser()
ap.add_argument("table")
ap.add_argument("--wide-readers", default="", help="comma-separated kernel names whose FETCH_SIZE gets the x2 correction (16-byte-per-lane streaming reads)")
ap.add_argument("--pairs-per-launch", type=float, default=32)
ap.add_argument("--note", default="")
ap.add_argument("-o", "--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json"))
a = ap.parse_args()
if a.wide_readers:
    WIDE_READERS = set(a.wide_readers.split(","))
out, valu = {}, {}
for r in csv.DictReader(open(a.table)):
    k = NAMES.get(r["kernel"])
    if k is None or not r.get("FETCH_SIZE_per_launch"):
        continue
    # gfx950 correction (see _correction): only for kernels whose reads are 16 B per lane; since round 2 no kernel of the hot path
    # streams dwordx4 from memory any more (descriptors are assembled in LDS from 4-byte plane words)
    kb = (2.0 if r["kernel"] in WIDE_READERS else 1.0) * float(r["FETCH_SIZE_per_launch"]) + float(r["WRITE_SIZE_per_launch"])
    out[k] = out.get(k, 0) + int(round(kb * 1024 / a.pairs_per_launch))
    if r.get("SQ_INSTS_VALU_per_launch"):
        valu[k] = valu.get(k, 0) + int(round(float(r["SQ_INSTS_VALU_per_launch"]) / a.pairs_per_launch))
json.dump({"_source": "%s (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, %s: %g pairs per launch, no kernel overlap)" % (a.table, a.note, a.pairs_per_launch),
           "_correction": "bytes = (c*FETCH_SIZE + WRITE_SIZE) * 1024 with c = 2 for kernels that stream 16 bytes per lane (on gfx950 FETCH_SIZE reports half of those bytes: MI355X_MICROARCH.md, HBM section; %s) and c = 1 for the others (4-byte-per-lane reads)" % (",".join(sorted(WIDE_READERS)) or "none in this build"),
           "bytes_per_pair": out,
           "_valu": "SQ_INSTS_VALU of the same run (its own --pmc pass): wave64 VALU instructions per pair; tools/valu_rate.hip prices them at 2 (add/and/mov/fma) to 4 (sad/min/max/med3/three-operand) cycles of one of the 1024 SIMDs",
           "valu_wave_insts_per_pair": valu}, open(a.out, "w"), indent=1)
print(json.dumps(out))
