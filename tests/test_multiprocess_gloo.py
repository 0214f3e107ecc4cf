"""CPU: the N>1 path (one process per GPU, batch sharded by rank, barrier + max-over-ranks timing, optional gather)
rehearsed with world_size 2 on the gloo backend."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    par = util.pkg("parallel")
    synth = util.pkg("synth")
    assert par.world() == (rank, world_size)
    B, H, W, D = 3, 40, 64, 16
    seeds = par.pair_seeds(rank, B)
    batch = np.stack([np.stack(synth.make_pair(s, H, W, D)) for s in seeds])
    # stand-in for the engine (no GPU here): a per-pair reduction that depends on the pair's bytes
    maps = torch.from_numpy(batch[:, 0].astype(np.float32) - batch[:, 1].astype(np.float32))
    dist.barrier()
    tmax = par.max_over_ranks(0.5 + rank)
    total = par.sum_over_ranks(float(B))
    gathered = par.gather_maps(maps, dst=0)
    lo, hi = par.shard_range(7, rank, world_size)
    out.put((rank, seeds, tmax, total, None if gathered is None else gathered.numpy(), (lo, hi), maps.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_timing_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, n0, g0, sh0, m0), (r1, s1, t1, n1, g1, sh1, m1) = res
    assert set(s0).isdisjoint(s1) and len(s0) == len(s1) == 3          # weak scaling: distinct pairs on every rank
    assert t0 == t1 == 1.5                                             # max over ranks
    assert n0 == n1 == 6.0                                             # whole-job unit count
    assert g1 is None and g0.shape == (6, 40, 64)
    assert np.array_equal(g0[:3], m0) and np.array_equal(g0[3:], m1)   # rank order == pair order
    assert sh0 == (0, 4) and sh1 == (4, 7)                             # strong-scaling split covers 7 units exactly once
