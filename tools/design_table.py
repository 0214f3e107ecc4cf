#!/usr/bin/env python3
"""DESIGN.md section 4's kernel table, regenerated from a committed profile:

    python tools/design_table.py profiles/r04f_serial_kernel_stats_pmc.csv [--pairs-per-launch 32]

Columns: median launch duration of the no-overlap pass (rocprofv3 --kernel-trace, >= 16 launches per kernel), HBM traffic from the
FETCH_SIZE / WRITE_SIZE passes (KB per launch as rocprofv3 reports them, no correction: no kernel of the path streams 16 bytes per lane),
wave64 VALU instructions from the SQ_INSTS_VALU pass, all per pair; SURVEY.md section 8(d)'s algorithmic bytes beside them."""
import argparse
import csv

N = 1242 * 375
GROUPS = [  # (name in the bench line, rocprof kernel names, what it replaces, SURVEY 8(d) bytes per pair)
    ("descriptor", ["k_sobel"], "`sobel3x3` (gradient planes; `createDescriptor`'s gather happens in LDS inside the matching kernels)", 2 * N),
    ("support_match", ["k_support"], "`computeMatchingDisparity` x2 + L/R test", 2 * N),
    ("support_filter", ["k_filter_classify", "k_filter_resolve", "k_filter_vertical", "k_filter_horizontal", "k_filter_collect", "k_filter_corners"],
     "`removeInconsistent/RedundantSupportPoints`, collection, `addCornerSupportPoints` (6 launches)", 0),
    ("delaunay_gpu", ["dg::k_delaunay_resident", "dg::k_delaunay_blob", "dg::k_dg_prepare_large_blob", "dg::k_dgl_subtrees_blob", "dg::k_dgl_top_blob"],
     "Triangle \"zQB\": sort, duplicate scan, alternating cuts, divide-and-conquer (the GPU's share of the chunks)", 0),
    ("grid_mark + grid_dilate", ["k_grid_mark", "k_grid_dilate"], "`createGrid`", 0),
    ("plane_fit", ["k_planes"], "`computeDisparityPlanes` + edge lines + tile binning", 0),
    ("triangles_raster", ["k_raster_tiles", "k_raster"], "scan conversion of `computeDisparity`", 0),
    ("dense_match", ["k_dense"], "`findMatch`, both sides", 10 * N),
    ("lr_check", ["k_lr2", "k_lr"], "`leftRightConsistencyCheck`", 12 * N),
    ("ccl_band + ccl_finish", ["k_ccl_band", "k_ccl_border", "k_ccl_total", "k_ccl_apply", "k_ccl_slow"], "`removeSmallSegments`", 16 * N),
    ("gap_rows + gap_cols", ["k_gap_rows", "k_gap_cols"], "`gapInterpolation`", 16 * N),
    ("adaptive_mean", ["k_amean", "k_amean_sub"], "`adaptiveMean`", 16 * N),
    ("median", ["k_median"], "`median`", 16 * N),
]

ap = argparse.ArgumentParser()
ap.add_argument("table")
ap.add_argument("--pairs-per-launch", type=float, default=32)
a = ap.parse_args()
rows = {r["kernel"]: r for r in csv.DictReader(open(a.table))}
print("| kernel | replaces | 8(d) MB/pair | HBM MB/pair | median us per %d-pair launch | us/pair | VALU wave-instructions/pair |" % a.pairs_per_launch)
print("|---|---|---|---|---|---|---|")
tot = [0.0, 0.0, 0.0, 0.0]
for name, kernels, what, alg in GROUPS:
    us = mb = vi = 0.0
    parts = []
    for k in kernels:
        r = rows.get(k)
        if not r:
            continue
        m = float(r.get("median_us") or r["avg_us"])
        us += m
        parts.append("%.1f" % m)
        mb += (float(r.get("FETCH_SIZE_per_launch") or 0) + float(r.get("WRITE_SIZE_per_launch") or 0)) * 1024 / a.pairs_per_launch / 1e6
        vi += float(r.get("SQ_INSTS_VALU_per_launch") or 0) / a.pairs_per_launch
    if not parts:
        continue
    tot[0] += alg / 1e6
    tot[1] += mb
    tot[2] += us / a.pairs_per_launch
    tot[3] += vi
    print("| `%s` | %s | %s | %.2f | %s | %.2f | %.2f M |" % (name, what, ("%.2f" % (alg / 1e6)) if alg else "-", mb, " + ".join(parts), us / a.pairs_per_launch, vi / 1e6))
print("| sum | | %.1f (88 N + the Sobel kernel's 2 N) | %.1f | | %.1f | %.2f M |" % (tot[0], tot[1], tot[2], tot[3] / 1e6))
