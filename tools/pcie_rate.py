#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-memory entry points (sv_process_batch_host / sv_submit_batch_host / sv_elas_process):
SURVEY.md section 8(d)'s pair = gray L+R in host memory -> D1 back in host memory.  Never the bench `value`.

    python tools/pcie_rate.py [--batch 256] [--steps 6] [--workload kitti_d128]
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"


def throughput(eng, e, L, R, d1, d2, steps):
    """`steps` batches submitted back to back (streamed), all outputs to the same arrays; pairs/s."""
    e.process_host(L, R, want_d2=d2 is not None, d1=d1, d2=d2)
    t0 = time.perf_counter()
    for _ in range(steps):
        e.submit_host(L, R, d1, d2)
    e.wait()
    return L.shape[0] * steps / (time.perf_counter() - t0)


def latency(eng, W, H, params, L1, R1, pinned, calls=200):
    alloc = eng.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
    L, R = alloc((H, W), np.uint8), alloc((H, W), np.uint8)
    Hm, Wm = (H // 2, W // 2) if params.subsampling else (H, W)
    D1, D2 = alloc((Hm, Wm), np.float32), alloc((Hm, Wm), np.float32)
    L[:], R[:] = L1, R1
    dims = (ctypes.c_int32 * 3)(W, H, W)
    e = eng.StereoEngine(W, H, params, n_workers=4, chunk=1, n_streams=1, n_slots=2)
    f = eng.lib().sv_elas_process
    args = (e._h, L.ctypes.data, R.ctypes.data, D1.ctypes.data, D2.ctypes.data, dims)
    for _ in range(20):
        assert f(*args) == 0
    ts = []
    for _ in range(calls):
        a = time.perf_counter()
        f(*args)
        ts.append(time.perf_counter() - a)
    e.close()
    return {"median": round(1e3 * float(np.median(ts)), 3), "p99": round(1e3 * float(np.percentile(ts, 99)), 3), "calls": calls}


def measure(eng, synth, W, H, D, B, steps, batch=None, scale=1, seed0=1000, lat_pair=None, subsampling=False):
    params = eng.SvParams.driver(D - 1)
    params.subsampling = 1 if subsampling else 0
    if batch is None:
        batch = synth.make_batch(seed0, B, H, W, D, scale=scale)
    Hm, Wm = (H // 2, W // 2) if subsampling else (H, W)
    out = {}
    e = eng.StereoEngine(W, H, params)
    for kind in ("pinned", "pageable"):
        alloc = eng.pinned_array if kind == "pinned" else (lambda shape, dt: np.zeros(shape, dt))
        L, R = alloc((B, H, W), np.uint8), alloc((B, H, W), np.uint8)
        L[:], R[:] = batch[:, 0], batch[:, 1]
        d1, d2 = alloc((B, Hm, Wm), np.float32), alloc((B, Hm, Wm), np.float32)
        r1 = throughput(eng, e, L, R, d1, None, steps)
        r2 = throughput(eng, e, L, R, d1, d2, steps)
        in_b, map_b = 2 * W * H, 4 * Wm * Hm
        out[kind] = {"pairs_per_s_d1": round(r1, 1), "pairs_per_s_d1_d2": round(r2, 1),
                     "h2d_GBps_d1": round(r1 * in_b / 1e9, 2), "d2h_GBps_d1": round(r1 * map_b / 1e9, 2),
                     "h2d_GBps_d1_d2": round(r2 * in_b / 1e9, 2), "d2h_GBps_d1_d2": round(r2 * 2 * map_b / 1e9, 2)}
    e.close()
    if lat_pair is None:
        lat_pair = (batch[0, 0], batch[0, 1])
    out["latency_ms_batch1_host"] = {k: latency(eng, W, H, params, lat_pair[0], lat_pair[1], k == "pinned") for k in ("pinned", "pageable")}
    out["bytes_per_pair"] = {"in": 2 * W * H, "d1": 4 * Wm * Hm}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--disp", type=int, default=128)
    args = ap.parse_args()
    eng = importlib.import_module(PKG + ".engine")
    synth = importlib.import_module(PKG + ".synth")
    lat_pair = None
    try:
        from PIL import Image
        gl = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
        gr = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
        if gl.shape == (args.height, args.width):
            lat_pair = (gl, gr)
    except (OSError, ImportError):
        pass
    print(json.dumps(measure(eng, synth, args.width, args.height, args.disp, args.batch, args.steps, lat_pair=lat_pair)))


if __name__ == "__main__":
    main()
