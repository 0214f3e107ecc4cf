#!/usr/bin/env python3
"""Per-kernel average launch durations (HIP events, serial configuration: one slot, one stream) - quick A/B of kernel variants
without rocprof:   SV_LIB_PATH=abl_tmp/lib_x.so python tools/ktime.py [--only dense,support]"""
import argparse
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--chunk", type=int, default=32)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--only", default="")
ap.add_argument("--real", action="store_true", help="copies of the committed kitti_mini pair 0 instead of synthetic pairs")
a = ap.parse_args()
eng = importlib.import_module(PKG + ".engine")
synth = importlib.import_module(PKG + ".synth")
W, H, D = 1242, 375, a.disp
b = synth.make_batch(1000, a.batch, H, W, D)
if a.real:
    from PIL import Image
    b[:, 0] = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
    b[:, 1] = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
left = torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda()
right = torch.from_numpy(np.ascontiguousarray(b[:, 1])).cuda()
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=a.chunk, n_streams=1, n_slots=1)
e.process_device(left, right)
e.timing(True)
for _ in range(a.reps):
    e.process_device(left, right)
kt = e.kernel_times()
only = [x for x in a.only.split(",") if x]
tot = 0.0
out = []
for k, (ms, calls) in sorted(kt.items(), key=lambda x: -x[1][0]):
    if not calls or k.startswith("host:"):
        continue
    us_pair = 1e3 * ms / (a.reps * a.batch)
    tot += us_pair
    if not only or any(o in k for o in only):
        out.append("%s %.1f us/launch %.2f us/pair" % (k, 1e3 * ms / calls, us_pair))
print(os.environ.get("SV_LIB_PATH", "default"), "| total %.1f us/pair |" % tot, " | ".join(out))
e.close()
