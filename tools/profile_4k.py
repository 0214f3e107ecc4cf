#!/usr/bin/env python3
"""One 4K chunk (3840 x 2160, D = 192) through a handle, for rocprofv3 --kernel-trace --stats:
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4k -- python3 tools/profile_4k.py [--triangulation gpu] [--workers 2]"""
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
ap.add_argument("--pairs", type=int, default=16)
ap.add_argument("--chunk", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--workers", type=int, default=2)
ap.add_argument("--triangulation", default="gpu")
ap.add_argument("--overlap", action="store_true")
a = ap.parse_args()
eng = importlib.import_module(PKG + ".engine")
synth = importlib.import_module(PKG + ".synth")
W, H, D = 3840, 2160, 192
b = synth.make_batch(5000, min(a.pairs, 4), H, W, D, scale=3)
b = np.concatenate([b] * (-(-a.pairs // b.shape[0])))[:a.pairs]
left = torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda()
right = torch.from_numpy(np.ascontiguousarray(b[:, 1])).cuda()
kw = {} if a.overlap else dict(n_streams=1, n_slots=1)
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=a.chunk, n_workers=a.workers, triangulation=a.triangulation, **kw)
print(e.query())
for _ in range(a.reps):
    d1, d2 = e.process_device(left, right)
torch.cuda.synchronize()
print("fallbacks", e.gpu_triangulation_fallbacks())
e.close()
