#!/usr/bin/env python3
"""Soak run of the single-pair path: the seven committed real frames and synthetic pairs, one pair per call on chunk-1 handles with every
latency_split policy, device and page-locked host memory, back to back and with pauses (the pollers go to sleep and come back); every call's
maps must equal a throughput handle's maps of the same pair.   SECONDS=60 python tools/soak_latency.py"""
import ctypes
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import util

pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
eng = importlib.import_module(pkg + ".engine")
SECONDS = float(os.environ.get("SECONDS", "60"))
H, W, D = 375, 1242, 128
p = eng.SvParams.driver(D - 1)
pairs = []
for i in (0, 3, 7, 10, 13, 17, 20):
    try:
        pairs.append((util.load_png("kitti%d_left.png" % i), util.load_png("kitti%d_right.png" % i)))
    except Exception:
        pass
syn = util.pkg("synth").make_batch(4242, 6, H, W, D)
pairs += [(syn[i, 0], syn[i, 1]) for i in range(6)]
ref = eng.StereoEngine(W, H, p, chunk=4, n_slots=2)
want = []
for a, b in pairs:
    L, R = torch.from_numpy(np.ascontiguousarray(a[None])).cuda(), torch.from_numpy(np.ascontiguousarray(b[None])).cuda()
    d1, d2 = ref.process_device(L, R)
    want.append((d1.clone(), d2.clone()))
ref.close()
dev = [(torch.from_numpy(np.ascontiguousarray(a[None])).cuda(), torch.from_numpy(np.ascontiguousarray(b[None])).cuda()) for a, b in pairs]
hl, hr = eng.pinned_array((H, W), np.uint8), eng.pinned_array((H, W), np.uint8)
h1, h2 = eng.pinned_array((H, W), np.float32), eng.pinned_array((H, W), np.float32)
dims = (ctypes.c_int32 * 3)(W, H, W)
rng = np.random.default_rng(3)
t_end = time.time() + SECONDS
calls = bad = 0
rounds = 0
while time.time() < t_end:
    split = rounds % 4
    workers = int(rng.choice([2, 4, 7, 14]))
    e = eng.StereoEngine(W, H, p, chunk=1, n_slots=2, n_streams=1, n_workers=workers, latency_split=split)
    o1, o2 = torch.empty((1, H, W), dtype=torch.float32, device="cuda"), torch.empty((1, H, W), dtype=torch.float32, device="cuda")
    t_round = time.time() + min(5.0, SECONDS / 8)
    while time.time() < t_round:
        i = int(rng.integers(0, len(pairs)))
        if rng.random() < 0.3:  # host pointers, page-locked
            hl[:], hr[:] = pairs[i]
            assert eng.lib().sv_elas_process(e._h, hl.ctypes.data, hr.ctypes.data, h1.ctypes.data, h2.ctypes.data, dims) == 0
            ok = np.array_equal(h1, want[i][0][0].cpu().numpy()) and np.array_equal(h2, want[i][1][0].cpu().numpy())
        else:
            e.process_device(dev[i][0], dev[i][1], o1, o2)
            ok = torch.equal(o1, want[i][0]) and torch.equal(o2, want[i][1])
        bad += 0 if ok else 1
        calls += 1
        r = rng.random()
        if r < 0.02:
            time.sleep(0.03)  # the pollers run out of patience
        elif r < 0.1:
            time.sleep(0.004)
    q = e.query()
    e.close()
    rounds += 1
    print("round %d: policy %d, %d pool threads -> %d thread(s) per triangulation; %d calls so far, %d wrong" % (rounds, split, workers, {0: 1, 1: 2, 2: 4}[q["latency_split"]], calls, bad), flush=True)
print("soak done: %d calls, %d wrong" % (calls, bad))
sys.exit(1 if bad else 0)
