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
