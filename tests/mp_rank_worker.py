"""One rank of the two-process GPU rehearsal (tests/test_multiprocess_gpu.py): a fresh process whose first GPU call happens
here, driving the real engine on its shard of the batch; control-plane collectives over gloo (both ranks share the one GPU
of the test box - the real runs use one GPU per rank and nccl = RCCL)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import util  # noqa: E402


def main():
    out_path = sys.argv[1]
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng, par, synth = util.pkg("engine"), util.pkg("parallel"), util.pkg("synth")
    dev = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    B, H, W, D = 5, 120, 320, 64
    seeds = par.pair_seeds(rank, B, seed0=300)
    batch = np.stack([np.stack(synth.make_pair(s, H, W, D)) for s in seeds])
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), device=dev, chunk=2, n_slots=3)
    info = e.query()
    try:
        left, right = torch.from_numpy(batch[:, 0].copy()).cuda(), torch.from_numpy(batch[:, 1].copy()).cuda()
        dist.barrier()
        d1, d2 = e.process_device(left, right)
        torch.cuda.synchronize()
        # host-memory entry on the same handle as well (what a caller without device tensors uses)
        h1, h2, st = e.process_host(batch[:, 0], batch[:, 1])
    finally:
        e.close()
    assert np.array_equal(h1, d1.cpu().numpy()) and np.array_equal(h2, d2.cpu().numpy()) and (st >= 3).all()
    tmax = par.max_over_ranks(1.0 + rank)
    total = par.sum_over_ranks(float(B))
    g1 = par.gather_maps(d1.cpu(), dst=0)
    g2 = par.gather_maps(d2.cpu(), dst=0)
    if rank == 0:
        np.savez(out_path, d1=g1.numpy(), d2=g2.numpy(), tmax=tmax, total=total, host_threads=info["host_threads"])
    else:
        np.savez(out_path + ".rank1.npz", host_threads=info["host_threads"], tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
