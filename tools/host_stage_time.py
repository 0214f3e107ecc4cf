#!/usr/bin/env python3
"""The host stage of a single frame (latency mode), piece by piece on this host's CPU: lattice filters (sv_host_support_filter) and the
Delaunay triangulation of the left / right vertex sets (whole: sv_host_delaunay; halves on two threads: sv_host_delaunay_par), for the
candidate lattice of kitti_mini pair 0.   python tools/host_stage_time.py [--disp 128]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
import util
pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
eng = importlib.import_module(pkg + ".engine")
l, r = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
H, W = l.shape
p = eng.SvParams.driver(a.disp - 1)
e = eng.StereoEngine(W, H, p, keep_debug=True)
e.process_device(torch.from_numpy(l[None].copy()).cuda(), torch.from_numpy(r[None].copy()).cuda())
dims = e.debug("dcan_dims").astype(np.int64)
dcan = e.debug("dcan_raw").reshape(int(dims[1]), int(dims[0])) if dims[0] * dims[1] == e.debug("dcan_raw").size else e.debug("dcan_raw")
sup = e.debug("support").reshape(-1, 3)
e.close()


def timed(f, *args, **kw):
    f(*args, **kw)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        f(*args, **kw)
    return (time.perf_counter() - t0) / a.reps * 1e6


print("lattice %s, %d support points" % (tuple(dcan.shape), sup.shape[0]))
print("lattice filters + corner points: %.1f us" % timed(eng.host_support_filter, p, dcan, W, H))
left = np.ascontiguousarray(sup[:, :2])
right = np.ascontiguousarray(np.stack([sup[:, 0] - sup[:, 2], sup[:, 1]], 1))
for name, xy in (("left", left), ("right", right)):
    print("%s set: order only %.1f us; triangulation %.1f us; halves on two threads %.1f us; quarters on four %.1f us" % (
        name, timed(eng.host_kd_order, xy), timed(eng.host_delaunay, xy), timed(eng.host_delaunay, xy, split=True, depth=1), timed(eng.host_delaunay, xy, split=True, depth=2)))
