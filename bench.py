#!/usr/bin/env python3
"""Benchmark of the MI355X-native ELAS stereo hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (gray pair -> final left disparity map + L/R-checked right map) over one
batch of KITTI-shaped stereo pairs that is already resident in HBM.  Metric = BASELINE.json's "stereo pairs/sec" at
1242x375, D=128 (throughput configuration, batch 256 per GPU = kitti_mini pair 0 + 255 synthetic pairs, SURVEY.md
section 8d config 3); the batch-1 latency ("ms/frame") is reported beside it.  One process per GPU (torch.distributed /
RCCL for the barrier and the max-over-ranks time only: pairs are independent, there is no data-path collective), weak
scaling.

Also on the JSON line:
  roofline        the dominant kernel (largest total time over ALL kernels in a pass without kernel overlap), its
                  algorithmic bytes as SURVEY.md section 8(d) counts them (dense matching = 2N read + 8N write) over its average
                  launch duration by HIP events inside the timed region, against the 8 TB/s HBM3E peak; the same
                  with the no-overlap duration (`serial`), with the interface bytes (`frac_interface`), and the
                  committed rocprofv3 summaries the durations can be checked against (`profile`)
  roofline_valu   what actually binds the matching kernels: VALU issue.  Wave instructions per pair (committed
                  SQ_INSTS_VALU pass), the issue floor they imply, and the SAD byte rate against the chip's
                  157 T byte-absdiff/s (live candidate counters of the no-overlap pass)
  host_to_host    SURVEY.md section 8(d)'s pair: gray L+R in host memory -> maps back in host memory, streamed
                  (sv_submit_batch_host), page-locked and pageable caller memory; never `value`
  cpu_baseline    the reference's own serial LIBELAS (oracle/_ref, compiled from /root/reference in the build
                  container) or, if that artefact is absent, our CPU restatement, timed on this host, 1 thread;
                  cpu_baseline_all_cores: one pair per process on every core this process may use
"""
import argparse
import csv
import ctypes
import glob
import importlib
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the first HIP call: one hardware queue per engine stream (engine.py explains)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"
sys.path.insert(0, ROOT)

W, H, D = 1242, 375, 128  # headline workload (BASELINE.json configs[1]/[2]); --workload switches to the other configs
WORKLOADS = {  # name: (W, H, D, default pairs per GPU per step, chunk, slots, synth scale, first seed, distinct pairs generated)
    "kitti_d128": (1242, 375, 128, 256, 0, 0, 1, 1000, 256),
    "kitti_d256": (1242, 375, 256, 64, 0, 0, 1, 1000, 64),      # configs[3]: LDS-pressure configuration
    "4k_d192": (3840, 2160, 192, 128, 0, 0, 3, 5000, 16),        # configs[4]: 128 pairs per GPU; 16 distinct pairs repeated (a 4K pair takes ~1 s to synthesise)
}
METRIC = {
    "kitti_d128": "stereo pairs/sec, KITTI 1242x375 D=128 (ms/frame at batch 1 in latency_ms_batch1)",
    "kitti_d256": "stereo pairs/sec, KITTI 1242x375 D=256 (LDS-pressure configuration; ms/frame at batch 1 in latency_ms_batch1)",
    "4k_d192": "stereo pairs/sec, synthetic 4K 3840x2160 D=192 (ms/frame at batch 1 in latency_ms_batch1)",
}
LATENCY_WORKERS = 7  # latency handles: the calling thread + 7 pool threads build the two triangulations as eight quarters
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# integer byte-absdiff peak: 1024 SIMDs x 64 lanes x 4 bytes per v_sad_u8, one wave instruction per 4 cycles (tools/valu_rate.hip), 2.4 GHz
SAD_PEAK_BYTE_OPS = 1024 * 64 * 4 / 4 * 2.4e9
KERNEL_TRACE_NAMES = {"descriptor": ["k_descriptor", "k_sobel"], "support_match": ["k_support"], "support_filter": ["k_filter_classify", "k_filter_resolve", "k_filter_vertical", "k_filter_collect"], "grid_mark": ["k_grid_mark"],
                      "grid_dilate": ["k_grid_dilate"], "plane_fit": ["k_planes"], "triangles_raster": ["k_raster_tiles"], "triangles_raster_fallback": ["k_raster"],
                      "dense_match": ["k_dense"], "lr_check": ["k_lr", "k_lr2"], "delaunay_gpu": ["dg::k_delaunay_blob"], "ccl_band": ["k_ccl_band"],
                      "ccl_finish": ["k_ccl_border", "k_ccl_total", "k_ccl_apply", "k_ccl_slow", "k_ccl_merge"], "gap_rows": ["k_gap_rows"], "gap_cols": ["k_gap_cols"],
                      "adaptive_mean": ["k_amean", "k_amean_sub"], "median": ["k_median"], "output": ["k_output"]}


def algorithmic_bytes_8d(N):
    """stage-minimum HBM bytes per pair, one read of each stage's inputs + one write of its outputs, no
    credit for redundant passes or materialised intermediates (descriptors, triangle ids, labels beyond the 8N the survey
    grants).  N = W*H of the map the stage works on.  Sum = 88N + the 2N the descriptor kernel's own gray read adds."""
    return {
        "descriptor": 2 * N,                 # k_sobel: gray L,R in (the gradient planes it writes are an intermediate: no credit)
        "support_match": 2 * N,              # "support 2N (gray L,R)"
        "dense_match": 2 * N + 8 * N,        # "dense 2N read + 8N write (D1,D2 f32)"
        "lr_check": 8 * N + 4 * N,           # "LR 8N read + 4N write"
        "ccl_band": 4 * N + 4 * N + 8 * N, "ccl_finish": 0,   # "speckle 4N + 4N + 8N labels" (charged to the band kernel)
        "gap_rows": 8 * N, "gap_cols": 8 * N,
        "adaptive_mean": 16 * N,             # H 8N + V 8N (one fused kernel)
        "median": 16 * N,                    # H 8N + V 8N (one fused kernel)
    }


def interface_bytes_per_pair(N, Wc, Hc, Wimg, MW, ncell):
    """HBM bytes each kernel must move for ONE pair given the buffers it is handed (du/dv gradient planes of 2 B/px per image,
    triangle ids, int16 WTA maps): what an ideal implementation of THIS kernel interface would move.  Reported as frac_interface."""
    grad = 2 * N  # du + dv byte planes of one image
    return {
        "descriptor": 2 * N + 2 * grad,                      # k_sobel: gray L,R in, gradient planes out
        "support_match": 2 * grad + 2 * Wc * Hc,             # both images' planes (every row is used by some lattice row), lattice out
        "support_filter": 2 * Wc * Hc + 12 * 2200,
        "grid_mark": 0, "grid_dilate": 2 * 2 * ncell * MW * 4, "plane_fit": 0,
        "triangles_raster": 2 * 4 * N, "triangles_raster_fallback": 0,
        "dense_match": 2 * grad + 2 * 4 * N + 2 * 2 * N,     # planes, triangle ids in; int16 WTA maps out
        "lr_check": 2 * 2 * N + 2 * 4 * N,
        "delaunay_gpu": 2 * (16 * 2200 + 12 * 4400),
        "ccl_band": 4 * N, "ccl_finish": 0,
        "gap_rows": 8 * N, "gap_cols": 8 * N, "adaptive_mean": 8 * N, "median": 8 * N + 4 * N,
        "output": 2 * 8 * N,
    }


def usable_cpus():
    """Cores this process may use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_CPU_WORKER = r"""
import os, sys, time
sys.path.insert(0, %(oracle)r); sys.path.insert(0, %(root)r)
import importlib, pyoracle
synth = importlib.import_module(%(pkg)r + ".synth")
lib = pyoracle.RefElas() if pyoracle.RefElas.available() else pyoracle.Oracle()
p = pyoracle.ElasParams.driver(%(D)d - 1); p.subsampling = %(sub)d
pairs = [synth.make_pair(s, %(H)d, %(W)d, %(D)d, scale=%(scale)d) for s in %(seeds)r]
print("ready", flush=True); sys.stdin.readline()
t0 = time.perf_counter()
for L, R in pairs: lib.process(p, L, R, canonical=False, reps=1)
print("done %%f" %% (time.perf_counter() - t0), flush=True)
"""


def cpu_baseline(sample_pairs, synth, subsampling=False, scale=1, all_cores_pairs=0):
    """Reference serial path (or our port of it) on this host's cores, bounded sample: 1 thread, and - one pair per process,
    the reference keeps global state (Triangle's LCG seed) - every core this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    if pyoracle.RefElas.available():
        lib, kind = pyoracle.RefElas(), "reference"
    else:
        lib, kind = pyoracle.Oracle(), "port"
    p = pyoracle.ElasParams.driver(D - 1)
    p.subsampling = 1 if subsampling else 0
    t = 0.0
    for i in range(sample_pairs):
        L, R = synth.make_pair(1000 + i, H, W, D, scale=scale)
        _, _, sec = lib.process(p, L, R, canonical=False, reps=1)
        t += sec
    one = {"value": round(sample_pairs / t, 3), "unit": "pairs/s", "cores": 1, "kind": kind,
           "sample": "%d synthetic pairs (seeds 1000..%d), %dx%d, D=%d, Elas::process only, %.1f s" % (sample_pairs, 1000 + sample_pairs - 1, W, H, D, t),
           "ms_per_pair": round(1e3 * t / sample_pairs, 2), "host_cpus": os.cpu_count()}
    many = None
    T = usable_cpus()
    if all_cores_pairs > 0 and T > 1:
        per = max(1, all_cores_pairs // T)
        procs = []
        for r in range(T):
            code = _CPU_WORKER % {"oracle": os.path.join(ROOT, "oracle"), "root": ROOT, "pkg": PKG, "D": D, "H": H, "W": W, "scale": scale,
                                  "sub": 1 if subsampling else 0, "seeds": [1000 + r * per + i for i in range(per)]}
            procs.append(subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        try:
            for pr in procs:
                assert pr.stdout.readline().startswith("ready")
            t0 = time.perf_counter()
            for pr in procs:
                pr.stdin.write("go\n")
                pr.stdin.flush()
            for pr in procs:
                assert pr.stdout.readline().startswith("done")
            wall = time.perf_counter() - t0
            many = {"value": round(T * per / wall, 3), "unit": "pairs/s", "cores": T, "kind": kind,
                    "sample": "%d processes x %d synthetic pairs each, one pair at a time per process, %.1f s wall" % (T, per, wall)}
        finally:
            for pr in procs:
                try:
                    pr.stdin.close()
                except OSError:
                    pass
                pr.wait(timeout=60)
    return one, many


def host_throughput(eng, e, batch, Hm, Wm, steps, pinned, want_d2):
    """`steps` host-memory batches submitted back to back (sv_submit_batch_host) and waited for: pairs/s, PCIe inclusive."""
    B = batch.shape[0]
    alloc = eng.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
    L, R = alloc((B, H, W), np.uint8), alloc((B, H, W), np.uint8)
    L[:], R[:] = batch[:, 0], batch[:, 1]
    d1 = alloc((B, Hm, Wm), np.float32)
    d2 = alloc((B, Hm, Wm), np.float32) if want_d2 else None
    e.process_host(L, R, want_d2=want_d2, d1=d1, d2=d2)  # warm-up (allocates the staging buffers at the first call)
    t0 = time.perf_counter()
    for _ in range(steps):
        e.submit_host(L, R, d1, d2)
    e.wait()
    return B * steps / (time.perf_counter() - t0)


def host_latency(eng, params, l1, r1, pinned, calls=200):
    """sv_elas_process (the reference's seam, host pointers) on a latency-mode handle: ms per call."""
    alloc = eng.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
    Hm, Wm = (H // 2, W // 2) if params.subsampling else (H, W)
    L, R, D1, D2 = alloc((H, W), np.uint8), alloc((H, W), np.uint8), alloc((Hm, Wm), np.float32), alloc((Hm, Wm), np.float32)
    L[:], R[:] = l1, r1
    dims = (ctypes.c_int32 * 3)(W, H, W)
    e = eng.StereoEngine(W, H, params, n_workers=LATENCY_WORKERS, chunk=1, n_streams=1, n_slots=2)
    f = eng.lib().sv_elas_process
    args = (e._h, L.ctypes.data, R.ctypes.data, D1.ctypes.data, D2.ctypes.data, dims)
    try:
        for _ in range(20):
            assert f(*args) == 0, eng.lib().sv_last_error(e._h)
        ts = []
        for _ in range(calls):
            a = time.perf_counter()
            f(*args)
            ts.append(time.perf_counter() - a)
    finally:
        e.close()
    return {"median": round(1e3 * float(np.median(ts)), 3), "p99": round(1e3 * float(np.percentile(ts, 99)), 3), "calls": calls}


def pcie_ceiling(mb=256, reps=5):
    """Raw DMA rates of this box between page-locked host memory and HBM (torch copies on two streams): the ceiling the
    host-to-host rates have to be read against."""
    import torch
    n = mb << 20
    hp, hq = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    dp, dq = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    dp.copy_(hp, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    hq.copy_(dq, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9

    run(True, True)
    return {"h2d_alone": round(run(True, False), 2), "d2h_alone": round(run(False, True), 2), "each_direction_when_both_run": round(run(True, True), 2),
            "how": "%d MB page-locked <-> HBM copies, %d per direction" % (mb, reps)}


def host_to_host(eng, e, params, batch, steps, lat_pair):
    Hm, Wm = (H // 2, W // 2) if params.subsampling else (H, W)
    in_b, map_b = 2 * W * H, 4 * Wm * Hm
    out = {"definition": "gray L+R u8 in host memory -> f32 maps back in host memory, %d pairs per batch, %d batches streamed (sv_submit_batch_host)" % (batch.shape[0], steps),
           "bytes_per_pair": {"in": in_b, "d1": map_b}, "pcie_ceiling_GBps": pcie_ceiling()}
    for kind in ("pinned", "pageable"):
        r1 = host_throughput(eng, e, batch, Hm, Wm, steps, kind == "pinned", False)
        r2 = host_throughput(eng, e, batch, Hm, Wm, steps, kind == "pinned", True)
        out[kind] = {"pairs_per_s_d1": round(r1, 1), "pairs_per_s_d1_d2": round(r2, 1),
                     "pcie_GBps_d1": {"h2d": round(r1 * in_b / 1e9, 2), "d2h": round(r1 * map_b / 1e9, 2)},
                     "pcie_GBps_d1_d2": {"h2d": round(r2 * in_b / 1e9, 2), "d2h": round(r2 * 2 * map_b / 1e9, 2)}}
    out["latency_ms_batch1_host"] = {k: host_latency(eng, params, lat_pair[0], lat_pair[1], k == "pinned") for k in ("pinned", "pageable")}
    return out


def profile_durations(kernel, pattern):
    """Average launch duration of `kernel` in the newest committed rocprofv3 summary matching `pattern` (profiles/*.csv, written
    by tools/summarize_rocprof.py); None when there is none."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    f = files[-1]
    tot_us, calls, ppl = 0.0, 0, None
    try:
        for r in csv.DictReader(open(f)):
            if r["kernel"] in KERNEL_TRACE_NAMES.get(kernel, []):
                tot_us += float(r["avg_us"]) * int(r["calls"])
                calls = max(calls, int(r["calls"]))
                if float(r["avg_us"]) > 0 and float(r.get("us_per_pair") or 0) > 0:
                    ppl = round(float(r["avg_us"]) / float(r["us_per_pair"]))
    except (OSError, KeyError, ValueError):
        return None
    if calls == 0:
        return None
    return {"file": os.path.relpath(f, ROOT), "avg_launch_us": round(tot_us / calls, 2), "pairs_per_launch": ppl}


def load_real_pair():
    try:
        from PIL import Image
        gl = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
        gr = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
        if gl.shape == (H, W):
            return np.ascontiguousarray(gl), np.ascontiguousarray(gr)
    except (OSError, ImportError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = the workload's default: 256 for the headline)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kitti_d128", help="kitti_d128 is BASELINE.json's metric configuration")
    ap.add_argument("--workers", type=int, default=0, help="host pool threads (0 = library default)")
    ap.add_argument("--streams", type=int, default=0, help="phase-2 HIP streams (0 = library default)")
    ap.add_argument("--slots", type=int, default=0, help="pipeline slots = chunks in flight (0 = library default)")
    ap.add_argument("--chunk", type=int, default=0, help="pairs per launch (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP events around every kernel launch of the timed region (1-2 %% of the rate) "
                    "instead of the dominant kernel only")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurements (profiling runs: keeps the per-kernel averages to the timed region's launches)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-to-host measurements")
    ap.add_argument("--no-real", action="store_true", help="skip the extra throughput measurement on copies of kitti_mini pair 0")
    ap.add_argument("--sync-steps", action="store_true", help="wait for each step before submitting the next (default: streamed submission)")
    ap.add_argument("--synthetic-only", action="store_true", help="all pairs synthetic (default: kitti_mini pair 0 + synthetic pairs, SURVEY.md 8d config 3)")
    ap.add_argument("--real-pair", action="store_true", help="fill the whole batch with copies of the committed kitti_mini pair 0")
    ap.add_argument("--subsampling", action="store_true", help="Elas::parameters::subsampling (the reference's s1 benchmark rows): half-resolution maps")
    ap.add_argument("--gather", action="store_true", help="after the timed region, also gather all left maps on rank 0 (RCCL) and report the time")
    args = ap.parse_args()
    global W, H, D
    W, H, D, wb, wchunk, wslots, wscale, wseed, wdistinct = WORKLOADS[args.workload]
    args.batch = args.batch or wb
    args.chunk = args.chunk or wchunk
    args.slots = args.slots or wslots

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    # BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks (ranks then share devices and
    # the control-plane collectives run on CPU tensors); the real runs use nccl (= RCCL) with one GPU per rank
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    eng = importlib.import_module(PKG + ".engine")
    synth = importlib.import_module(PKG + ".synth")
    par = importlib.import_module(PKG + ".parallel")
    B = args.batch
    # weak scaling: every rank owns B pairs of its own (seeds wseed + rank*B + i); no data-path collective
    seeds = par.pair_seeds(rank, B, seed0=wseed)
    distinct = min(B, wdistinct)
    batch = synth.make_batch(seeds[0], distinct, H, W, D, scale=wscale)
    if distinct < B:
        batch = np.concatenate([batch] * (-(-B // distinct)))[:B]
    real = load_real_pair()
    data_desc = "synthetic (%d distinct pairs%s)" % (distinct, ", repeated to %d" % B if distinct < B else "")
    if args.real_pair:
        assert real is not None, "the committed pair is 1242x375"
        batch[:, 0], batch[:, 1] = real
        data_desc = "kitti_mini pair 0 replicated"
    elif real is not None and not args.synthetic_only:
        batch[0, 0], batch[0, 1] = real  # "pair 0 plus 255 synthetic"
        data_desc = "kitti_mini pair 0 (committed gray fixture) + %d synthetic pairs (seeds %d..%d)" % (B - 1, seeds[1], seeds[-1])
    left = torch.from_numpy(np.ascontiguousarray(batch[:, 0])).cuda()
    right = torch.from_numpy(np.ascontiguousarray(batch[:, 1])).cuda()
    params = eng.SvParams.driver(D - 1)
    params.subsampling = 1 if args.subsampling else 0  # the reference's "s1" rows (results_log.txt): half-resolution maps
    Hm, Wm = (H // 2, W // 2) if args.subsampling else (H, W)
    d1 = torch.empty((B, Hm, Wm), dtype=torch.float32, device="cuda")
    d2 = torch.empty((B, Hm, Wm), dtype=torch.float32, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- pass without kernel overlap (one slot, one stream; every kernel timed): clean per-kernel durations, the dominant
    # kernel by total time over ALL kernels, and the live candidate counters of the matching kernels
    serial_k, counters, dom = None, None, None
    if not args.no_kernel_timing:
        schunk = args.chunk or (32 if W < 2000 else 4)
        es = eng.StereoEngine(W, H, params, device=local_rank, n_workers=args.workers, chunk=schunk, n_streams=1, n_slots=1)
        nb = min(B, 2 * schunk)
        es.process_device(left[:nb], right[:nb], d1[:nb], d2[:nb])
        es.timing(True)
        es.process_device(left[:nb], right[:nb], d1[:nb], d2[:nb])
        kt = es.kernel_times()
        es.timing(False)
        nc = min(nb, 8)  # candidate counters: separate (slower) instantiations of the matching kernels, a few pairs suffice
        es.counters(True)
        es.process_device(left[:nc], right[:nc], d1[:nc], d2[:nc])
        counters = {k: v / nc for k, v in es.counters().items()}
        es.close()
        serial_k = {k: (v[0], v[1], nb * 1.0 / max(v[1], 1)) for k, v in kt.items() if v[1] > 0 and not k.startswith("host:")}  # (total ms, launches, pairs per launch)
        dom = max(serial_k, key=lambda k: serial_k[k][0])

    engine = eng.StereoEngine(W, H, params, device=local_rank, n_workers=args.workers, chunk=args.chunk, n_streams=args.streams, n_slots=args.slots)
    engine_info = engine.query()
    for _ in range(args.warmup):
        engine.process_device(left, right, d1, d2)
    if not args.no_kernel_timing:  # HIP events on the engine's own streams, inside the timed region
        engine.timing(True, only=None if args.time_all_kernels else (dom,))
    barrier()
    t0 = time.perf_counter()
    if args.sync_steps:
        for _ in range(args.steps):
            engine.process_device(left, right, d1, d2)
    else:  # "streamed": the K batches are handed to the engine back to back, the next one fills the pipeline while the last drains
        for _ in range(args.steps):
            engine.submit_device(left, right, d1, d2)
        engine.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    ktimes = engine.kernel_times() if not args.no_kernel_timing else {}
    engine.timing(False)
    engine_info["gpu_triangulation_share"] = engine.gpu_triangulation_share()  # host mode: what the dispatcher's load balancing handed to the GPU kernel
    elapsed = par.max_over_ranks(elapsed, device=coll_dev)
    gather_ms = None
    if args.gather and world > 1:  # the optional "trivial gather" of finished maps on rank 0 over RCCL/xGMI, outside the timed region
        barrier()
        g0 = time.perf_counter()
        allmaps = par.gather_maps(d1 if backend == "nccl" else d1.cpu(), dst=0)
        barrier()
        gather_ms = round(1e3 * (time.perf_counter() - g0), 3)
        del allmaps
    valid_frac = float((d1 >= 0).float().mean().item())
    checksum = float(d1.double().sum().item())

    # ---- the same engine on a batch of copies of the real pair (real maps are far more fragmented than synthetic ones)
    real_rate = None
    if rank == 0 and world == 1 and real is not None and not args.no_real and not args.real_pair:
        rl = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(real[0], (B, H, W)))).cuda()
        rr = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(real[1], (B, H, W)))).cuda()
        engine.process_device(rl, rr, d1, d2)
        rsteps = max(3, args.steps // 2)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(rsteps):
            engine.submit_device(rl, rr, d1, d2)
        engine.wait()
        torch.cuda.synchronize()
        real_rate = {"value": round(B * rsteps / (time.perf_counter() - r0), 1), "unit": "pairs/s", "data": "kitti_mini pair 0 x %d" % B, "steps": rsteps}
        del rl, rr

    # ---- host memory in / out through the same engine (PCIe inclusive)
    h2h = None
    lat_pair = real if real is not None else (batch[0, 0], batch[0, 1])
    if rank == 0 and world == 1 and not args.no_host:
        h2h = host_to_host(eng, engine, params, batch, max(3, args.steps // 2), lat_pair)
        if args.no_latency:
            h2h.pop("latency_ms_batch1_host", None)

    # batch-1 latency on rank 0 (ms/frame), SURVEY.md §8d config 2: pair 0 of kitti_mini (the committed gray fixture; the first
    # pair of the batch if the fixture is absent), one pair per call, 200 timed calls after 20 warm-ups, device memory in and out
    lat_ms = None
    if rank == 0 and not args.no_latency:
        e1 = eng.StereoEngine(W, H, params, device=local_rank, n_workers=LATENCY_WORKERS, chunk=1, n_streams=1, n_slots=2)
        l1 = torch.from_numpy(np.array(lat_pair[0][None])).cuda()
        r1 = torch.from_numpy(np.array(lat_pair[1][None])).cuda()
        which = "kitti_mini pair 0" if real is not None else "first pair of the batch"
        o1, o2 = d1[:1].clone(), d2[:1].clone()
        for _ in range(20):
            e1.process_device(l1, r1, o1, o2)
        ts = []
        for _ in range(200):
            torch.cuda.synchronize()
            a = time.perf_counter()
            e1.process_device(l1, r1, o1, o2)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - a)
        e1.close()
        lat_ms = {"median": round(1e3 * float(np.median(ts)), 3), "p99": round(1e3 * float(np.percentile(ts, 99)), 3), "pair": which, "calls": 200,
                  "memory": "device in / device out (host memory: host_to_host.latency_ms_batch1_host)"}
    engine.close()

    if rank == 0:
        total_pairs = B * world * args.steps
        rate = total_pairs / elapsed
        out = {
            "metric": METRIC[args.workload] + (" subsampling=1" if args.subsampling else ""),
            "value": round(rate, 2),
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": data_desc,
            "config": {"workload": "%s_%dx%d_D%d_batch%d_per_gpu_streamed" % (args.workload.split("_")[0], W, H, D, B), "width": W, "height": H, "disp_max": D - 1,
                       "preset": "driver(MIDDLEBURY+only_left+adaptive_mean+median)" + ("+subsampling" if args.subsampling else ""), "pairs_per_gpu_per_step": B,
                       "parallelism": "batch-sharded x%d, no data-path collective" % world, "engine": engine_info},
            "latency_ms_batch1": lat_ms, "value_real_pair": real_rate, "host_to_host": h2h, "gather_ms": gather_ms,
            "valid_fraction": round(valid_frac, 4), "checksum_rank0": checksum,
        }
        if h2h:
            out["value_host_to_host"] = h2h["pinned"]["pairs_per_s_d1"]
        if ktimes and serial_k:
            N = W * H
            step = params.candidate_stepsize
            Wc, Hc = (W + step - 1) // step, (H + step - 1) // step
            gw, gh = -(-W // params.grid_size), -(-H // params.grid_size)
            alg = algorithmic_bytes_8d(Wm * Hm if args.subsampling else N)
            alg["descriptor"], alg["support_match"] = 2 * N, 2 * N
            itf = interface_bytes_per_pair(N, Wc, Hc, W, (D + 31) // 32, gw * gh)
            host = {k: v for k, v in ktimes.items() if k.startswith("host:")}
            ktimes = {k: v for k, v in ktimes.items() if not k.startswith("host:") and v[1] > 0}
            out["host_stage_cpu_ms_per_pair"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in host.items()}
            # dominant kernel: argmax of total time over ALL kernels of the no-overlap pass
            s_ms, s_calls, s_ppl = serial_k[dom]
            ms, calls = ktimes[dom]
            ppl = B * args.steps / calls  # rank 0's launches of this kernel each cover one chunk
            avg_s, s_avg_s = 1e-3 * ms / calls, 1e-3 * s_ms / s_calls
            a8 = alg.get(dom, 0)
            achieved = a8 * ppl / avg_s / 1e9
            s_achieved = a8 * s_ppl / s_avg_s / 1e9
            traffic, traffic_src, pmc = None, None, {}
            try:  # HBM bytes from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as the microarch guide prescribes)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                if dom in pmc["bytes_per_pair"]:
                    traffic = int(pmc["bytes_per_pair"][dom] * ppl)
                    traffic_src = pmc["_source"]
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {
                "bound": "hbm", "kernel": dom, "dominant_by_time": dom,
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_pair": a8, "algorithmic_bytes_per_launch": int(a8 * ppl), "bytes_definition": "SURVEY.md 8(d) stage minimum (dense = 2N read + 8N write); no credit for intermediates (gradient planes, triangle ids)",
                "avg_launch_us": round(1e6 * avg_s, 2), "pairs_per_launch": ppl, "duration_source": "HIP events on the launching stream, timed region (kernels of other streams overlap)",
                "serial": {"avg_launch_us": round(1e6 * s_avg_s, 2), "pairs_per_launch": s_ppl, "achieved": round(s_achieved, 2), "frac": round(s_achieved / HBM_PEAK_GBS, 5),
                           "duration_source": "HIP events, one slot / one stream pass of this run (no kernel overlap)"},
                "frac_interface": round(itf.get(dom, 0) * ppl / avg_s / 1e9 / HBM_PEAK_GBS, 5), "interface_bytes_per_pair": itf.get(dom, 0),
                "profile": {"pipelined": profile_durations(dom, "r*_bench_pipelined_kernel_stats.csv"), "serial": profile_durations(dom, "r*_serial_kernel_stats_pmc.csv")},
                "whole_pipeline": {"bytes_per_pair_8d": 88 * N, "achieved": round(88 * N * (rate / world) / 1e9, 2), "frac": round(88 * N * (rate / world) / 1e9 / HBM_PEAK_GBS, 5)},
            }
            # VALU issue: what binds the matching kernels
            vi = pmc.get("valu_wave_insts_per_pair", {})
            valu = {"note": "integer VALU issue, not HBM, binds the matching kernels: 1024 SIMDs, 2.4 GHz, 4 cycles per wave64 v_sad_u8 / min / max / med3 (tools/valu_rate.hip)",
                    "sad_peak_byte_ops_per_s": SAD_PEAK_BYTE_OPS, "wave_insts_source": pmc.get("_source"), "kernels": {}}
            byte_ops = {"dense_match": 16.0 * (counters or {}).get("dense_candidates", 0.0), "support_match": 64.0 * (counters or {}).get("support_energies", 0.0)}
            for kk in ("dense_match", "support_match", "descriptor"):
                if kk not in serial_k:
                    continue
                k_ms, k_calls, k_ppl = serial_k[kk]
                us_pair = 1e3 * k_ms / (k_calls * k_ppl)
                ent = {"serial_us_per_pair": round(us_pair, 3)}
                if vi.get(kk):
                    ent["wave_insts_per_pair"] = vi[kk]
                    ent["issue_floor_us_per_pair_at_4_cycles"] = round(vi[kk] * 4 / (1024 * 2.4e9) * 1e6, 3)
                    ent["issue_floor_us_per_pair_at_2_cycles"] = round(vi[kk] * 2 / (1024 * 2.4e9) * 1e6, 3)
                if byte_ops.get(kk):
                    ent["sad_byte_ops_per_pair"] = int(byte_ops[kk])
                    ent["sad_byte_ops_per_s"] = round(byte_ops[kk] / (us_pair * 1e-6), 1)
                    ent["sad_frac_of_peak"] = round(byte_ops[kk] / (us_pair * 1e-6) / SAD_PEAK_BYTE_OPS, 4)
                valu["kernels"][kk] = ent
            if counters:
                valu["counters_per_pair"] = {k: round(v, 1) for k, v in counters.items()}
            out["roofline_valu"] = valu
            out["roofline_by_kernel"] = {kk: {"serial_us_per_pair": round(1e3 * v[0] / (v[1] * v[2]), 3),
                                               "frac_8d_serial": round(alg.get(kk, 0) / (1e-3 * v[0] / (v[1] * v[2])) / 1e9 / HBM_PEAK_GBS, 5),
                                               "frac_interface_serial": round(itf.get(kk, 0) / (1e-3 * v[0] / (v[1] * v[2])) / 1e9 / HBM_PEAK_GBS, 5)}
                                         for kk, v in sorted(serial_k.items(), key=lambda kv: -kv[1][0])}
            out["serial_kernel_us_per_pair_sum"] = round(sum(1e3 * v[0] / (v[1] * v[2]) for v in serial_k.values()), 3)
            out["kernel_ms_per_pair_timed_region"] = {k: round(v[0] / (B * args.steps), 5) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])}
        if world == 1 and args.cpu_sample > 0:
            one, many = cpu_baseline(args.cpu_sample if W < 2000 else min(args.cpu_sample, 4), synth, args.subsampling, scale=wscale,
                                     all_cores_pairs=(4 * usable_cpus() if W < 2000 else 0))
            out["cpu_baseline"] = one
            if many:
                out["cpu_baseline_all_cores"] = many
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
