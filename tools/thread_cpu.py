#!/usr/bin/env python3
"""Where a rank's host CPU time goes: the headline batch streamed for a few seconds, CPU seconds per thread (utime + stime of
/proc/self/task/*/stat, thread names from comm: the engine names its own threads sv-*) per wall second.

    python tools/thread_cpu.py [--workers 1] [--seconds 4] [--triangulation gpu]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def snapshot():
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (name, int(f[11]) / tick, int(f[12]) / tick)  # utime, stime
        except (OSError, ValueError):
            pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--triangulation", default="auto")
    ap.add_argument("--event-sync", default=None)
    ap.add_argument("--trace", action="store_true", help="the issuer's wall-clock split (lat_trace), printed when the engine closes")
    ap.add_argument("--cpus", type=int, default=0, help="restrict the process to this many CPUs first")
    a = ap.parse_args()
    if a.cpus:
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:a.cpus])
    import importlib
    import numpy as np
    import torch
    pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
    eng = importlib.import_module(pkg + ".engine")
    synth = importlib.import_module(pkg + ".synth")
    W, H, D = 1242, 375, 128
    batch = synth.make_batch(1000, 32, H, W, D)
    batch = np.concatenate([batch] * (a.batch // 32))
    left = torch.from_numpy(np.ascontiguousarray(batch[:, 0])).cuda()
    right = torch.from_numpy(np.ascontiguousarray(batch[:, 1])).cuda()
    d1 = torch.empty((a.batch, H, W), dtype=torch.float32, device="cuda")
    d2 = torch.empty_like(d1)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=a.workers, triangulation=a.triangulation, event_sync=a.event_sync)
    if a.trace:
        e.debug_set("lat_trace", 1)
    for _ in range(3):
        e.process_device(left, right, d1, d2)
    torch.cuda.synchronize()
    s0, t0, n = snapshot(), time.perf_counter(), 0
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(8):
            e.submit_device(left, right, d1, d2)
        n += 8
        e.wait()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    s1 = snapshot()
    print("%.0f pairs/s, engine %s" % (n * a.batch / wall, e.query()))
    rows = {}
    for tid, (name, u, s) in s1.items():
        u0, st0 = s0.get(tid, (name, 0.0, 0.0))[1:]
        r = rows.setdefault(name, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += u - u0
        r[2] += s - st0
    print("%-18s %4s %8s %8s   (cores busy: user, system)" % ("thread", "n", "user", "sys"))
    for name, (cnt, u, s) in sorted(rows.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        if u + s > 0.005 * wall:
            print("%-18s %4d %8.3f %8.3f" % (name, cnt, u / wall, s / wall))
    print("%-18s %4s %8.3f" % ("total", "", sum(u + s for _, u, s in rows.values()) / wall))
    e.close()


if __name__ == "__main__":
    main()
