"""One rank of the launcher / chunked-gather rehearsal (tests/test_launcher.py): started by launcher.spawn_ranks with the env://
rendezvous variables set, gloo backend, CPU tensors.  Rank 0 prints one JSON line (what the launcher relays)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import util  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "gather"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "sleep":  # the launcher's signal test: report the PID, then wait to be stopped
        print("pid %d" % os.getpid(), file=sys.stderr, flush=True)
        import time
        time.sleep(300)
        sys.exit(0)
    if mode == "fail" and rank == 1:
        sys.exit(7)  # before the rendezvous: the other rank blocks in init_process_group until the launcher stops it
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = util.pkg("parallel")
    B, H, W, g = 7, 5, 9, 3  # a ragged last chunk
    maps = (torch.arange(B * H * W, dtype=torch.float32).reshape(B, H, W) + 1000.0 * rank)
    cg = par.ChunkedGather(B, H, W, torch.float32, g, torch.device("cpu"), dst=0, stage_on_cpu=True)
    handles = [cg.submit(maps, k) for k in range(cg.nchunks)]
    for hnd in handles:
        cg.wait(hnd)
    ok = None
    if rank == 0:
        ok = all(bool(torch.equal(cg.root[r], maps - 1000.0 * rank + 1000.0 * r)) for r in range(world))
    tmax = par.max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"ok": ok, "world": world, "nchunks": cg.nchunks, "root_shape": list(cg.root.shape), "tmax": tmax, "launcher": os.environ.get("SV_LAUNCHER"),
                          "local_world_size": os.environ.get("LOCAL_WORLD_SIZE")}), flush=True)
    else:
        print("rank %d done" % rank, flush=True)  # must NOT reach the launcher's stdout
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
