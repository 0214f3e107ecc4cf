#!/usr/bin/env python3
"""Zero-copy probe: page-locked host arrays handed to the DEVICE entry points (the kernels read the gray rows / write the maps over
PCIe themselves), against the staged host path of the same library on the same box.

    python tools/zc_probe.py [--reps 3] [--steps 20] [--mode in|out|both|none]
"""
import argparse
import ctypes
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--mode", default="both")
ap.add_argument("--workers", type=int, default=0)
a = ap.parse_args()
pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
eng = importlib.import_module(pkg + ".engine")
synth = importlib.import_module(pkg + ".synth")
W, H, D = 1242, 375, 128
batch = synth.make_batch(1000, 32, H, W, D)
batch = np.concatenate([batch] * (a.batch // 32))
L, R = eng.pinned_array((a.batch, H, W), np.uint8), eng.pinned_array((a.batch, H, W), np.uint8)
L[:], R[:] = batch[:, 0], batch[:, 1]
d1h = eng.pinned_array((a.batch, H, W), np.float32)
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=a.workers)
lib = eng.lib()
ref, _, _ = e.process_host(L, R, want_d2=False)
ref = ref.copy()
Ld, Rd = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
d1d = torch.zeros((a.batch, H, W), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
pin = Ld.data_ptr() if a.mode in ("out", "none") else L.ctypes.data
pin_r = Rd.data_ptr() if a.mode in ("out", "none") else R.ctypes.data
pout = d1d.data_ptr() if a.mode in ("in", "none") else d1h.ctypes.data


def sub():
    rc = lib.sv_submit_batch_device(e._h, pin, pin_r, a.batch, W, pout, None, None)
    assert rc == 0, rc


d1h[:] = 0
sub()
e.wait()
got = d1h if a.mode in ("out", "both") else d1d.cpu().numpy()
print("mode", a.mode, "maps equal the staged host path:", bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32))), flush=True)
rates = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sub()
    e.wait()
    rates.append(a.batch * a.steps / (time.perf_counter() - t0))
print("direct", a.mode, " ".join("%.0f" % r for r in rates), flush=True)
e.close()
