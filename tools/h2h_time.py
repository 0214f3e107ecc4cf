#!/usr/bin/env python3
"""Host-to-host rate of one library build (SURVEY 8(d)'s pair: gray L+R in page-locked host memory -> f32 D1 back), for A/B runs:

    SV_LIB_PATH=abl_tmp/lib_x.so python tools/h2h_time.py [--reps 3] [--steps 20] [--pageable] [--d2] [--dmap]
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch  # noqa: F401  (before the library: one libamdhip64)

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--pageable", action="store_true")
ap.add_argument("--d2", action="store_true")
ap.add_argument("--dmap", action="store_true")
ap.add_argument("--trace", action="store_true")
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--workers", type=int, default=0)
a = ap.parse_args()
pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
eng = importlib.import_module(pkg + ".engine")
synth = importlib.import_module(pkg + ".synth")
W, H, D = 1242, 375, 128
batch = synth.make_batch(1000, 32, H, W, D)
batch = np.concatenate([batch] * (a.batch // 32))
alloc = (lambda shape, dt: np.zeros(shape, dt)) if a.pageable else eng.pinned_array
L, R = alloc((a.batch, H, W), np.uint8), alloc((a.batch, H, W), np.uint8)
L[:], R[:] = batch[:, 0], batch[:, 1]
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_slots=a.slots, chunk=a.chunk, n_workers=a.workers)
if a.trace:
    e.debug_set("lat_trace", 1)
def throttled():
    """(periods throttled, ms throttled) of this cgroup so far: a pool that fills the CPU quota gets every thread stopped in bursts"""
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0)) / 1e3
    except (OSError, ValueError):
        return 0, 0.0


rates = []
if a.dmap:
    dm = alloc((a.batch, H, W), np.uint8)
    e.process_host_dmap(L, R, dmap=dm)
    sub = lambda: e.submit_host_dmap(L, R, dm)
else:
    d1 = alloc((a.batch, H, W), np.float32)
    d2 = alloc((a.batch, H, W), np.float32) if a.d2 else None
    e.process_host(L, R, want_d2=a.d2, d1=d1, d2=d2)
    sub = lambda: e.submit_host(L, R, d1, d2)
th0 = throttled()
c0 = time.process_time()
w0 = time.perf_counter()
for _ in range(a.reps):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sub()
    e.wait()
    rates.append(a.batch * a.steps / (time.perf_counter() - t0))
th1 = throttled()
print("cores busy %.1f, cgroup throttled %d periods / %.0f ms, host threads %d;" % ((time.process_time() - c0) / (time.perf_counter() - w0), th1[0] - th0[0], th1[1] - th0[1], e.query()["host_threads"]), end=" ")
print(os.environ.get("SV_LIB_PATH", "default"), "slots %d chunk %d" % (a.slots, a.chunk), "pageable" if a.pageable else "pinned", "dmap" if a.dmap else ("d1+d2" if a.d2 else "d1"), " ".join("%.0f" % r for r in rates), flush=True)
e.close()
