"""GPU (MI355X): the N > 1 path with the REAL engine - two fresh processes (one per rank, as torch.distributed.run starts
them; LOCAL_WORLD_SIZE=2) share the test box's GPU, shard the batch by rank, and gather their maps on rank 0.
Per-rank maps equal the oracle, the host pool is split between the ranks, the gathered maps are in rank order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util
from pyoracle import ElasParams

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_drive_the_engine(oracle, tmp_path):
    eng, par, synth = util.pkg("engine"), util.pkg("parallel"), util.pkg("synth")
    out = str(tmp_path / "rank0.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.HERE, "mp_rank_worker.py"), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append(p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    r0, r1 = np.load(out), np.load(out + ".rank1.npz")
    B, H, W, D = 5, 120, 320, 64
    assert r0["d1"].shape == (2 * B, H, W) and float(r0["tmax"]) == 2.0 == float(r1["tmax"]) and float(r0["total"]) == 2.0 * B
    po = ElasParams.driver(D - 1)
    for rank in range(2):
        for i, s in enumerate(par.pair_seeds(rank, B, seed0=300)):  # rank order == pair order in the gathered array
            L, R = synth.make_pair(s, H, W, D)
            o1, o2, _ = oracle.process(po, L, R)
            assert np.array_equal(r0["d1"][rank * B + i].view(np.uint8), o1.view(np.uint8)), (rank, i)
            assert np.array_equal(r0["d2"][rank * B + i].view(np.uint8), o2.view(np.uint8)), (rank, i)
    # the host pool is sized from the cores the process may use, shared between the ranks of the node (LOCAL_WORLD_SIZE)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=2, n_slots=2)
    alone = e.query()["host_threads"]
    e.close()
    sys.path.insert(0, util.ROOT)
    import bench
    usable = bench.usable_cpus()
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()[0] != "max"
    except OSError:
        quota = False

    def expected(share):  # engine.cpp:default_pool_size: two cores stay free under a CPU quota; at most 16 threads without one, 32 with
        share = max(1, share)
        return max(1, min(32 if quota else 16, share - 2 if quota and share > 4 else share))

    assert alone == expected(usable)
    assert int(r0["host_threads"]) == int(r1["host_threads"]) == expected(usable // 2)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): the parent spawns two fresh ranks, which share this
    box's GPU (BENCH_BACKEND=gloo: collectives on CPU tensors), pass the parity gate, run the timed region and the chunked,
    overlapped gather; rank 0's JSON line is the parent's only stdout."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env["BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--min-seconds", "0.5", "--batch", "128", "--no-host",
           "--no-latency", "--no-real", "--cpu-sample", "0", "--no-configs", "--gather"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=360)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["launcher"] == "self" and r["collective_backend"] == "gloo" and r["parity_gate"] == "pass"
    assert len(r["per_rank_pairs_per_s"]) == 2 and r["value"] > 0 and r["steps"] >= 3
    g = r["with_gather"]
    assert g["root_block0_equals_own_maps"] is True and g["chunk_pairs"] == 64 and g["pairs_per_s"] > 0
    g8 = r["with_gather_u8"]  # the same gather with the driver's 8-bit disparity images: a quarter of the bytes
    assert g8["root_block0_equals_own_maps"] is True and g8["bytes_into_root_per_step"] * 4 == g["bytes_into_root_per_step"]
    assert r["roofline"]["kernel"] and "cpu_baseline" not in r  # the CPU baseline is a rank-0, N=1 measurement
    # the line's tail validates an N-rank run by itself (the nccl = RCCL branch has the same keys; there `rccl_ranks` = N and the ranks'
    # devices must be distinct - here the two gloo ranks share the box's one GPU on purpose)
    t = r["trailer"]
    assert t["n_ranks"] == 2 and t["collective_backend"] == "gloo" and t["rccl_ranks"] is None and t["distinct_devices"] == 1
    assert len(t["per_rank_pairs_per_s"]) == 2 and abs(t["per_rank_mean_pairs_per_s"] - r["value"] / 2) < 1.0
    assert t["root_blocks_match_rank_checksums"] == {"headline": [True, True]}
