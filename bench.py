#!/usr/bin/env python3
"""Benchmark of the MI355X-native ELAS stereo hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (gray pair -> final left disparity map + L/R-checked right map) over one
batch of KITTI-shaped synthetic stereo pairs that is already resident in HBM.  Metric = BASELINE.json's
"stereo pairs/sec" at 1242x375, D=128 (throughput configuration, batch 256 per GPU); the batch-1 latency
("ms/frame") is reported beside it.  One process per GPU (torch.distributed / RCCL for the barrier and the
max-over-ranks time only: pairs are independent, there is no data-path collective), weak scaling.

Also on the JSON line:
  roofline      the dominant kernel's achieved algorithmic HBM bytes/s (HIP events on the launching streams,
                measured inside the timed region) against the 8 TB/s HBM3E peak
  cpu_baseline  the reference's own serial LIBELAS (oracle/_ref, compiled from /root/reference in the build
                container) or, if that artefact is absent, our CPU restatement, timed on this host, 1 thread
"""
import argparse
import importlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the first HIP call: one hardware queue per engine stream (engine.py explains)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"
sys.path.insert(0, ROOT)

W, H, D = 1242, 375, 128  # headline workload (BASELINE.json configs[1]/[2]); --workload switches to the other configs
WORKLOADS = {  # name: (W, H, D, default pairs per GPU per step, chunk, slots, synth scale, first seed)
    "kitti_d128": (1242, 375, 128, 256, 0, 0, 1, 1000),
    "kitti_d256": (1242, 375, 256, 64, 0, 0, 1, 1000),     # configs[3]: LDS-pressure configuration
    "4k_d192": (3840, 2160, 192, 32, 4, 4, 3, 5000),        # configs[4] shape (128 pairs per GPU there); 0.8 GB per pair in flight
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes_per_pair(N, Wc, Hc, Wimg, MW, ncell):
    """Minimal HBM bytes each kernel must move for ONE pair given its interface (one read of its inputs, one write of
    its outputs; no credit for re-reads).  N = W*H.  See DESIGN.md §Kernels."""
    desc = 16 * N
    return {
        "descriptor": 2 * N + 2 * desc,                       # gray L,R in; 16 B/px descriptors out (both images)
        "support_match": 2 * (2 * (Hc - 1)) * Wimg * 16 + 2 * Wc * Hc,  # descriptor rows v+-2 of every lattice row, both images; lattice out
        "support_filter": 2 * Wc * Hc + 12 * 2200,            # lattice in, ~2.2k support points out
        "grid_mark": 0, "grid_dilate": 2 * 2 * ncell * MW * 4,
        "plane_fit": 0,
        "triangles_raster": 2 * 4 * N, "triangles_raster_fallback": 0,                        # one tri_id write per covered pixel, both sides
        "dense_match": 2 * desc + 2 * 4 * N + 2 * 2 * N,      # both descriptor images, tri_id in, int16 WTA out (both sides)
        "lr_check": 2 * 2 * N + 2 * 4 * N,                    # int16 WTA maps in; checked left map + the caller's right map out
        "delaunay_gpu": 2 * (16 * 2200 + 12 * 4400),         # GPU triangulation mode: ~2.2k support points + vertex order in, ~4.4k triangles out, per side
        "ccl_band": 4 * N, "ccl_finish": 0,
        "gap_rows": 8 * N, "gap_cols": 8 * N, "adaptive_mean": 8 * N, "median": 8 * N + 4 * N,
        "output": 2 * 8 * N,
    }


def cpu_baseline(sample_pairs, synth, subsampling=False):
    """Reference serial path (or our port of it) on this host's cores, 1 thread, bounded sample."""
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
        L, R = synth.make_pair(1000 + i, H, W, D, scale=3 if W > 2000 else 1)
        _, _, sec = lib.process(p, L, R, canonical=False, reps=1)
        t += sec
    return {"value": round(sample_pairs / t, 3), "unit": "pairs/s", "cores": 1, "kind": kind,
            "sample": "%d synthetic pairs (seeds 1000..%d), %dx%d, D=%d, Elas::process only, %.1f s" % (sample_pairs, 1000 + sample_pairs - 1, W, H, D, t),
            "ms_per_pair": round(1e3 * t / sample_pairs, 2), "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU per step (0 = the workload's default: 256 for the headline)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kitti_d128", help="kitti_d128 is BASELINE.json's metric configuration")
    ap.add_argument("--workers", type=int, default=0, help="host pool threads (0 = library default)")
    ap.add_argument("--streams", type=int, default=0, help="phase-2 HIP streams (0 = library default)")
    ap.add_argument("--slots", type=int, default=0, help="pipeline slots = chunks in flight (0 = library default)")
    ap.add_argument("--chunk", type=int, default=0, help="pairs per launch (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="pairs timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP events around every kernel launch (the full per-kernel table; 1-2 %% of the rate) "
                    "instead of the three largest kernels only")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement (profiling runs: keeps the per-kernel averages to the timed region's launches)")
    ap.add_argument("--sync-steps", action="store_true", help="wait for each step before submitting the next (default: streamed submission)")
    ap.add_argument("--serial-kernels", action="store_true", help="extra pass with one slot / one stream (no kernel overlap) to get clean per-kernel times")
    ap.add_argument("--real-pair", action="store_true", help="fill the batch with copies of the committed kitti_mini pair 0 instead of synthetic pairs (sanity check of the synthetic workload: real maps are more fragmented)")
    ap.add_argument("--subsampling", action="store_true", help="Elas::parameters::subsampling (the reference's s1 benchmark rows): half-resolution maps")
    ap.add_argument("--gather", action="store_true", help="after the timed region, also gather all left maps on rank 0 (RCCL) and report the time")
    args = ap.parse_args()
    global W, H, D
    W, H, D, wb, wchunk, wslots, wscale, wseed = WORKLOADS[args.workload]
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
    # weak scaling: every rank owns B distinct pairs (seeds 1000 + rank*B + i); no data-path collective
    seeds = par.pair_seeds(rank, B, seed0=wseed)
    batch = synth.make_batch(seeds[0], B, H, W, D, scale=wscale)
    data_desc = "synthetic"
    if args.real_pair:
        from PIL import Image
        gl = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
        gr = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
        assert gl.shape == (H, W), "the committed pair is 1242x375"
        batch[:, 0], batch[:, 1] = gl, gr
        data_desc = "kitti_mini pair 0 replicated"
    left = torch.from_numpy(np.ascontiguousarray(batch[:, 0])).cuda()
    right = torch.from_numpy(np.ascontiguousarray(batch[:, 1])).cuda()
    params = eng.SvParams.driver(D - 1)
    params.subsampling = 1 if args.subsampling else 0  # the reference's "s1" rows (results_log.txt): half-resolution maps
    Hm, Wm = (H // 2, W // 2) if args.subsampling else (H, W)
    d1 = torch.empty((B, Hm, Wm), dtype=torch.float32, device="cuda")
    d2 = torch.empty((B, Hm, Wm), dtype=torch.float32, device="cuda")
    engine = eng.StereoEngine(W, H, params, device=local_rank, n_workers=args.workers, chunk=args.chunk, n_streams=args.streams, n_slots=args.slots)
    engine_info = engine.query()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        engine.process_device(left, right, d1, d2)
    if not args.no_kernel_timing:  # HIP events on the engine's own streams, inside the timed region
        engine.timing(True, only=None if args.time_all_kernels else ("dense_match", "support_match", "descriptor"))
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

    serial_k = None
    if args.serial_kernels and rank == 0:
        es = eng.StereoEngine(W, H, params, device=local_rank, n_workers=args.workers, chunk=args.chunk or 16, n_streams=1, n_slots=1)
        es.process_device(left, right, d1, d2)
        es.timing(True)
        es.process_device(left, right, d1, d2)
        kt = es.kernel_times()
        es.close()
        serial_k = {k: round(1e3 * v[0] / B, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]) if v[1] > 0 and not k.startswith("host:")}
        serial_k["_sum"] = round(sum(serial_k.values()), 3)

    # batch-1 latency on rank 0 (ms/frame), SURVEY.md §8d config 2: pair 0 of kitti_mini (the committed gray fixture; the first
    # synthetic pair if the fixture is absent), one pair per call, 200 timed calls after 20 warm-ups
    lat_ms = None
    if rank == 0 and not args.no_latency:
        e1 = eng.StereoEngine(W, H, params, device=local_rank, n_workers=4, chunk=1, n_streams=1, n_slots=2)
        l1, r1, which = left[:1].contiguous(), right[:1].contiguous(), "synthetic seed %d" % seeds[0]
        try:
            from PIL import Image
            gl = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_left.png")))
            gr = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "kitti0_right.png")))
            if gl.shape == (H, W):
                l1, r1, which = torch.from_numpy(np.array(gl[None])).cuda(), torch.from_numpy(np.array(gr[None])).cuda(), "kitti_mini pair 0"
        except (OSError, ImportError):
            pass
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
        lat_ms = {"median": round(1e3 * float(np.median(ts)), 3), "p99": round(1e3 * float(np.percentile(ts, 99)), 3), "pair": which, "calls": 200}
    engine.close()

    if rank == 0:
        total_pairs = B * world * args.steps
        out = {
            "metric": "stereo pairs/sec, KITTI 1242x375 D=128 (ms/frame at batch 1 in latency_ms_batch1)",
            "value": round(total_pairs / elapsed, 2),
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": data_desc,
            "config": {"workload": "%s_%dx%d_D%d_batch%d_per_gpu_streamed" % (args.workload.split("_")[0], W, H, D, B), "width": W, "height": H, "disp_max": D - 1,
                       "preset": "driver(MIDDLEBURY+only_left+adaptive_mean+median)" + ("+subsampling" if args.subsampling else ""), "pairs_per_gpu_per_step": B,
                       "parallelism": "batch-sharded x%d, no data-path collective" % world, "engine": engine_info},
            "latency_ms_batch1": lat_ms, "gather_ms": gather_ms, "serial_kernel_us_per_pair": serial_k,
            "valid_fraction": round(valid_frac, 4), "checksum_rank0": checksum,
        }
        if ktimes:
            N = W * H
            step = params.candidate_stepsize
            Wc, Hc = (W + step - 1) // step, (H + step - 1) // step
            gw, gh = -(-W // params.grid_size), -(-H // params.grid_size)
            alg = algorithmic_bytes_per_pair(N, Wc, Hc, W, (D + 31) // 32, gw * gh)
            host = {k: v for k, v in ktimes.items() if k.startswith("host:")}
            ktimes = {k: v for k, v in ktimes.items() if not k.startswith("host:")}
            out["host_stage_cpu_ms_per_pair"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in host.items()}
            tot = {k: v[0] for k, v in ktimes.items() if v[1] > 0}
            # the roofline kernel is the largest of the three kernels whose grids fill the chip; the event time of the small
            # latency-bound kernels (lattice filter, GPU triangulation, gap_cols, speckle passes) is mostly time spent starved
            # beside another stream's kernel, not work
            big = [kk for kk in ("dense_match", "support_match", "descriptor") if kk in tot]
            dom = max(big or [kk for kk in tot if kk not in ("support_filter", "delaunay_gpu")], key=tot.get)
            ms, calls = ktimes[dom]
            pairs_per_launch = B * args.steps / calls  # rank 0's launches of this kernel each cover one chunk
            avg_s = 1e-3 * ms / calls
            achieved = alg[dom] * pairs_per_launch / avg_s / 1e9
            traffic, traffic_src, valu = None, None, None
            try:  # HBM bytes from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as the microarch guide prescribes)
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    pmc = json.load(f)
                if dom in pmc["bytes_per_pair"]:
                    traffic = int(pmc["bytes_per_pair"][dom] * pairs_per_launch)
                    traffic_src = pmc["_source"]
                vi = pmc.get("valu_wave_insts_per_pair", {}).get(dom)
                if vi:  # what bounds the matching kernels: VALU issue; 1024 SIMDs, 2.4 GHz, 2..4 cycles per instruction
                    per_launch = vi * pairs_per_launch
                    valu = {"wave_insts_per_launch": int(per_launch), "issue_time_us_at_2_and_4_cycles": [round(per_launch * c / (1024 * 2.4e9) * 1e6, 1) for c in (2, 4)],
                            "source": "SQ_INSTS_VALU, profiles/pmc_traffic.json"}
            except (OSError, ValueError, KeyError):
                pass
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": int(alg[dom] * pairs_per_launch),
                               "avg_launch_us": round(1e6 * avg_s, 2), "pairs_per_launch": pairs_per_launch,
                               "algorithmic_bytes_per_pair": alg[dom], "valu": valu}
            out["roofline_by_kernel"] = {kk: {"achieved_GBps": round(alg.get(kk, 0) * (B * args.steps / ktimes[kk][1]) / (1e-3 * ktimes[kk][0] / ktimes[kk][1]) / 1e9, 1),
                                               "frac": round(alg.get(kk, 0) * (B * args.steps / ktimes[kk][1]) / (1e-3 * ktimes[kk][0] / ktimes[kk][1]) / 1e9 / HBM_PEAK_GBS, 4)}
                                         for kk in tot if alg.get(kk, 0) > 0}
            gpu_ms_total = sum(tot.values())
            out["kernel_ms_per_pair"] = {k: round(v / (B * args.steps), 5) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
            out["kernel_ms_per_pair"]["_sum"] = round(gpu_ms_total / (B * args.steps), 5)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample if W < 2000 else min(args.cpu_sample, 4), synth, args.subsampling)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
