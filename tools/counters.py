#!/usr/bin/env python3
"""Work counters of the matching kernels on a few pairs: python tools/counters.py [--real] [--disp 128]"""
import argparse, importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"
ap = argparse.ArgumentParser()
ap.add_argument("--real", action="store_true")
ap.add_argument("--disp", type=int, default=128)
ap.add_argument("--pairs", type=int, default=8)
a = ap.parse_args()
eng = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth")
W, H, D = 1242, 375, a.disp
b = synth.make_batch(1000, a.pairs, H, W, D)
if a.real:
    from PIL import Image
    b[:, 0] = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
    b[:, 1] = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=a.pairs, n_streams=1, n_slots=1)
e.counters(True)
e.process_device(torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda(), torch.from_numpy(np.ascontiguousarray(b[:, 1])).cuda())
print(json.dumps({k: v / a.pairs for k, v in e.counters().items()}))
e.close()
