#!/usr/bin/env python3
"""Runs the hot path a few times with NO kernel overlap (one slot, one stream) so that per-kernel rocprofv3 numbers
(durations, PMC counters) are clean.  Usage on the GPU box:

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 tools/profile_run.py
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc1 -- python3 tools/profile_run.py
"""
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
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--chunk", type=int, default=16)
ap.add_argument("--reps", type=int, default=9, help="passes over the batch: the first is a warm-up, 8 x (batch / chunk) = 16 timed launches per kernel by default")
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--overlap", action="store_true", help="use the default pipelined configuration instead")
a = ap.parse_args()
eng = importlib.import_module(PKG + ".engine")
synth = importlib.import_module(PKG + ".synth")
W, H, D = 1242, 375, a.disp
b = synth.make_batch(1000, a.batch, H, W, D)
left = torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda()
right = torch.from_numpy(np.ascontiguousarray(b[:, 1])).cuda()
if a.overlap:
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=a.chunk)
else:
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=a.chunk, n_streams=1, n_slots=1)
for _ in range(a.reps):
    d1, d2 = e.process_device(left, right)
torch.cuda.synchronize()
e.close()
print("done", float((d1 >= 0).float().mean()))
