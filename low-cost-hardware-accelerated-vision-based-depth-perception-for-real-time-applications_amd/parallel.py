"""Multi-GPU plumbing: one process per GPU, pairs sharded by rank, no collective on the data path.

Stereo pairs are independent units, so a batch of B pairs per rank needs no exchange at all.  What
torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is used for:
  * the barrier + max-over-ranks of the step time in bench.py,
  * optionally gathering the finished disparity maps on one rank (`gather_maps`), the "trivial gather" of the north star:
    over xGMI every sender has its own direct link into the root, so a plain gather (no ring) is the right collective.
"""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(total, rank, world_size):
    """Contiguous [lo, hi) slice of `total` units owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pair_seeds(rank, per_rank, seed0=1000):
    """Synthetic-pair seeds of a rank under weak scaling: every rank gets `per_rank` distinct pairs."""
    return [seed0 + rank * per_rank + i for i in range(per_rank)]


def max_over_ranks(seconds, device=None):
    rank, ws = world()
    if ws == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    rank, ws = world()
    if ws == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_maps(maps, dst=0):
    """Gathers per-rank [B, H, W] maps on `dst` -> [world*B, H, W] there, None elsewhere (rank order = pair order)."""
    rank, ws = world()
    if ws == 1:
        return maps
    bufs = [torch.empty_like(maps) for _ in range(ws)] if rank == dst else None
    dist.gather(maps.contiguous(), bufs, dst=dst)
    return torch.cat(bufs, 0) if rank == dst else None
