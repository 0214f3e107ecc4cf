#!/usr/bin/env python3
"""Where a single frame's time goes (batch 1, device memory): the handle's own wall-clock split (sv_debug_set "lat_trace"), printed when the
handle closes.   python tools/latency_trace.py [--disp 128] [--calls 300]"""
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
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("--workers", type=int, default=0)
ap.add_argument("--size", choices=["kitti", "4k"], default="kitti", help="kitti_mini pair 0 (1242x375) or a synthetic 3840x2160 pair")
ap.add_argument("--period-ms", type=float, default=0.0, help="a frame every so many milliseconds (a camera) instead of back to back")
ap.add_argument("--host", choices=["pinned", "pageable"], default=None, help="sv_elas_process with host pointers (the reference's seam) instead of device tensors")
a = ap.parse_args()
import util
pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
eng = importlib.import_module(pkg + ".engine")
l, r = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
if a.size == "4k":
    b4 = util.pkg("synth").make_batch(5000, 1, 2160, 3840, a.disp)
    l, r = b4[0, 0], b4[0, 1]
H, W = l.shape
if a.host:
    import ctypes
    alloc = eng.pinned_array if a.host == "pinned" else (lambda shape, dt: np.zeros(shape, dt))
    hl, hr, h1, h2 = alloc((H, W), np.uint8), alloc((H, W), np.uint8), alloc((H, W), np.float32), alloc((H, W), np.float32)
    hl[:], hr[:] = l, r
    dims = (ctypes.c_int32 * 3)(W, H, W)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(a.disp - 1), chunk=1, n_slots=2, n_streams=1, n_workers=a.workers)
    f = eng.lib().sv_elas_process
    args = (e._h, hl.ctypes.data, hr.ctypes.data, h1.ctypes.data, h2.ctypes.data, dims)
    for _ in range(20):
        assert f(*args) == 0
    e.debug_set("lat_trace", 1)
    ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        f(*args)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("D=%d, %s host memory: median %.3f ms, p99 %.3f ms over %d calls" % (a.disp, a.host, np.median(ts), np.percentile(ts, 99), a.calls), flush=True)
    e.close()
    sys.exit(0)
L, R = torch.from_numpy(l[None].copy()).cuda(), torch.from_numpy(r[None].copy()).cuda()
d1 = torch.empty((1, H, W), dtype=torch.float32, device="cuda")
d2 = torch.empty_like(d1)
e = eng.StereoEngine(W, H, eng.SvParams.driver(a.disp - 1), chunk=1, n_slots=2, n_streams=1, n_workers=a.workers)
for _ in range(20):
    e.process_device(L, R, d1, d2)
e.debug_set("lat_trace", 1)
ts = []
for _ in range(a.calls):
    t0 = time.perf_counter()
    e.process_device(L, R, d1, d2)
    ts.append(time.perf_counter() - t0)
    if a.period_ms > 0:
        time.sleep(max(0.0, a.period_ms * 1e-3 - (time.perf_counter() - t0)))
ts = np.array(ts) * 1e3
if a.period_ms > 0:
    print("one frame every %.1f ms:" % a.period_ms, end=" ")
print("D=%d: median %.3f ms, p99 %.3f ms over %d calls; engine %s" % (a.disp, np.median(ts), np.percentile(ts, 99), a.calls, e.query()), flush=True)
e.close()
