"""CPU: `python bench.py --gpus N` without an outside launcher - the parent starts N fresh ranks itself (launcher.py, standard
library only, never touches the GPU), relays rank 0's line and propagates failures; and the chunked gather the ranks run
(parallel.ChunkedGather) fills one preallocated [world, B, H, W] buffer on rank 0 in rank order.  gloo, world size 2 and 3."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

import util

WORKER = os.path.join(util.HERE, "mp_gather_worker.py")


@pytest.mark.parametrize("nranks", [2, 3, 8])  # 8: the ranks of one MI355X node
def test_launcher_runs_ranks_and_relays_rank0(nranks):
    launcher = util.pkg("launcher")
    out = io.StringIO()
    rc = launcher.spawn_ranks([sys.executable, WORKER, "gather"], nranks, out=out, timeout=240)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # rank 0's line only
    r = json.loads(lines[0])
    assert r == {"ok": True, "world": nranks, "nchunks": 3, "root_shape": [nranks, 7, 5, 9], "tmax": float(nranks), "launcher": "self", "local_world_size": str(nranks)}


def test_launcher_propagates_a_failing_rank_and_stops_the_others():
    launcher = util.pkg("launcher")
    t0 = time.time()
    rc = launcher.spawn_ranks([sys.executable, WORKER, "fail"], 2, out=io.StringIO(), timeout=240)
    assert rc == 7
    assert time.time() - t0 < 120  # rank 0 was stopped, not waited for until the rendezvous timed out


def test_launcher_stops_its_ranks_when_it_is_terminated():
    """A SIGTERM to the launcher (a harness timeout) must not leave the ranks behind: they are stopped, exit code 128 + 15."""
    import signal
    code = ("import sys, importlib; sys.path.insert(0, %r); l = importlib.import_module(%r + '.launcher'); "
            "sys.exit(l.spawn_ranks([sys.executable, %r, 'sleep'], 2, timeout=240))" % (util.ROOT, util.PKG, WORKER))
    p = subprocess.Popen([sys.executable, "-c", code], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    pids = []
    while len(pids) < 2:
        line = p.stderr.readline()
        assert line, "the ranks did not start"
        if line.startswith("pid "):
            pids.append(int(line.split()[1]))
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    for pid in pids:  # gone (reaped by the launcher before it exited)
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_host_share_restriction_before_any_gpu_call():
    """launcher.restrict_to_host_share: affinity = usable CPUs / share, LOCAL_WORLD_SIZE = share, without importing torch."""
    code = ("import os, sys, json, importlib; sys.path.insert(0, %r); l = importlib.import_module(%r + '.launcher'); before = l.usable_cpus(); "
            "kept = l.restrict_to_host_share(4); print(json.dumps([before, kept, sorted(os.sched_getaffinity(0)), os.environ['LOCAL_WORLD_SIZE'], 'torch' in sys.modules]))"
            % (util.ROOT, util.PKG))
    before, kept, now, lws, torch_loaded = json.loads(subprocess.check_output([sys.executable, "-c", code], text=True))
    assert len(kept) == max(1, before // 4) and kept == now and lws == "4" and torch_loaded is False


def test_host_share_slices_of_different_ranks_are_disjoint():
    """launcher.restrict_to_host_share(share, local_rank): rank r takes slice r + 1 (mod share) of the CPU list, so the ranks of a node get
    distinct CPUs and a single sampled rank 0 does not sit on the list's first CPUs (CPU 0: interrupts, housekeeping)."""
    code = ("import os, sys, json, importlib; sys.path.insert(0, %r); l = importlib.import_module(%r + '.launcher'); "
            "allowed = sorted(os.sched_getaffinity(0)); out = []\n"
            "for r in range(2):\n"
            "    os.sched_setaffinity(0, allowed); out.append(l.restrict_to_host_share(2, r))\n"
            "print(json.dumps([allowed, out]))" % (util.ROOT, util.PKG))
    allowed, kept = json.loads(subprocess.check_output([sys.executable, "-c", code], text=True))
    if len(allowed) < 4:
        pytest.skip("fewer than four CPUs")
    assert not set(kept[0]) & set(kept[1]) and len(kept[0]) == len(kept[1]) >= 1
    assert set(kept[0]) | set(kept[1]) <= set(allowed)


def test_pool_size_of_a_pinned_rank_without_a_cpu_quota():
    """A rank whose mask is already its share of the machine (restrict_to_host_share, torchrun ranks pinned per rank) keeps that share
    on a host without a CPU quota: the mask is not divided by LOCAL_WORLD_SIZE a second time (engine.cpp: host_cpu_share)."""
    ncpu = os.cpu_count()
    usable = sorted(os.sched_getaffinity(0))
    if ncpu < 4 or len(usable) < 2:
        pytest.skip("needs a mask of two CPUs on a machine with at least four")
    code = ("import os, sys, ctypes; os.sched_setaffinity(0, %r); os.environ['LOCAL_WORLD_SIZE'] = %r; "
            "L = ctypes.CDLL(%r); L.sv_default_host_threads.argtypes = [ctypes.c_int]; print(L.sv_default_host_threads(1))")
    lib = os.path.join(util.ROOT, util.PKG, "libstereo_vision_hip.so")
    ranks = ncpu // 2  # a 2-CPU mask is exactly one rank's share
    pinned = int(subprocess.check_output([sys.executable, "-c", code % (set(usable[:2]), str(ranks), lib)], text=True))
    assert pinned == 2
    # a mask wider than the share IS the node's budget and is divided between the ranks
    wide = int(subprocess.check_output([sys.executable, "-c", code % (set(usable), "2", lib)], text=True))
    assert wide == min(16, len(usable) // 2 if len(usable) * 2 > ncpu else len(usable))  # (16: the pool's cap without a quota)


def test_launcher_module_is_stdlib_only():
    """The parent of a multi-GPU run must not initialise HIP: the launcher imports neither torch nor the engine."""
    code = "import sys, importlib; importlib.import_module(%r + '.launcher'); print(int('torch' in sys.modules), int('numpy' in sys.modules))" % util.PKG
    out = subprocess.check_output([sys.executable, "-c", code], cwd=util.ROOT, text=True)
    assert out.split() == ["0", "0"]


def test_bench_without_launcher_starts_its_own_ranks():
    """No GPU here: both children stop at bench.py's 'needs a GPU' assertion - the point is that `--gpus 2` with WORLD_SIZE
    unset spawns ranks (each reports its own failure) and the parent returns non-zero instead of asserting on WORLD_SIZE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by tests/test_multiprocess_gpu.py::test_bench_launches_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert p.stderr.count("bench.py needs a GPU") >= 1 and "WORLD_SIZE=" not in p.stderr
    assert p.stdout.strip() == ""


def test_bench_tables_are_consistent():
    """bench.py's static tables: every workload has a parity-gate fixture in tests/golden/digests.json of its own size and disparity
    range, a gather chunk, and SURVEY 8(d)'s stage bytes add up to 88 N (+ 2 N for the Sobel kernel's own read of the gray pair)."""
    sys.path.insert(0, util.ROOT)
    import bench
    dig = util.digests()
    for name, (W, H, D, B, *_rest) in bench.WORKLOADS.items():
        e = dig[bench.GATES[name]]
        assert e["shape"] == [H, W] and e["disp_max"] == D - 1 and e["preset"] == "driver", name
        assert set(("final1", "final2")) <= set(e["stages"]) and 1 <= bench.GATHER_CHUNK[name] <= B
    N = 1242 * 375
    a = bench.algorithmic_bytes_8d(N)
    assert sum(a.values()) == 90 * N and a["dense_match"] == 10 * N
    assert "k_filter_horizontal" in bench.KERNEL_TRACE_NAMES["support_filter"] and "dg::k_dgl_top_blob" in bench.KERNEL_TRACE_NAMES["delaunay_gpu"]
