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


class ChunkedGather:
    """The "trivial gather" of the north star, chunked so that it overlaps the engine's kernels: every rank's
    [B, H, W] maps end up in ONE preallocated [world, B, H, W] buffer on `dst` (rank order = pair order; no torch.cat, no second
    copy), `chunk` pairs per collective.  A chunk is handed over as soon as the engine has finished it (`engine.wait_batches`),
    while later chunks are still being computed; over xGMI each of the world-1 senders has its own direct link into the root,
    so a plain gather (grouped send/recv under RCCL) is the right collective - no ring.

    stage_on_cpu: control-plane rehearsal on the gloo backend (CPU tensors; ranks may share a GPU)."""

    def __init__(self, B, H, W, dtype, chunk, device, dst=0, stage_on_cpu=False):
        self.rank, self.world = world()
        self.B, self.chunk, self.dst, self.stage_on_cpu = int(B), max(1, int(chunk)), dst, stage_on_cpu
        self.nchunks = -(-self.B // self.chunk)
        self.root = None
        if self.rank == dst:
            self.root = torch.empty((self.world, B, H, W), dtype=dtype, device="cpu" if stage_on_cpu else device)
        self.bytes_per_chunk_into_root = (self.world - 1) * self.chunk * H * W * torch.empty((), dtype=dtype).element_size()

    def submit(self, maps, k):
        """Starts the gather of chunk k (pairs [k*chunk, (k+1)*chunk) of every rank's `maps`); returns a handle for wait()."""
        lo, hi = k * self.chunk, min(self.B, (k + 1) * self.chunk)
        src = maps[lo:hi]
        if self.stage_on_cpu:
            src = src.cpu()
        if self.world == 1:
            self.root[0, lo:hi].copy_(src)
            return None, src
        outs = [self.root[r, lo:hi] for r in range(self.world)] if self.rank == self.dst else None
        return dist.gather(src, outs, dst=self.dst, async_op=True), src

    @staticmethod
    def wait(handle):
        work, _src = handle
        if work is not None:
            work.wait()


def gather_maps(maps, dst=0):
    """Gathers per-rank [B, H, W] maps on `dst` -> [world*B, H, W] there, None elsewhere (rank order = pair order)."""
    rank, ws = world()
    if ws == 1:
        return maps
    bufs = [torch.empty_like(maps) for _ in range(ws)] if rank == dst else None
    dist.gather(maps.contiguous(), bufs, dst=dst)
    return torch.cat(bufs, 0) if rank == dst else None
