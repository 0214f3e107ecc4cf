"""Starts one fresh process per GPU (rank) for a multi-GPU run that was invoked as a plain `python bench.py --gpus N`.

Standard library only, and it never touches the GPU: a process that has initialised HIP must not be re-executed (on this
pool that takes the machine down), so the parent stays a pure launcher - it spawns N children with the environment
torch.distributed's env:// rendezvous reads (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
relays rank 0's JSON line (its other stdout lines go to stderr) and returns the first non-zero exit code after stopping the other ranks.
`python -m torch.distributed.run ... bench.py` keeps working: then WORLD_SIZE is already set and no launcher runs.

The reference has no counterpart (one GPU, blocking copies: src/parallel_includes/elas/elas_gpu.cu:537-563).
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank, nranks, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), SV_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer-to-peer set-up needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def spawn_ranks(argv, nranks, out=None, err=None, timeout=None, poll=0.05):
    """Runs `argv` once per rank (ranks 0..nranks-1 of one node).  Rank 0's JSON lines go to `out` as they appear; its other stdout
    lines, the other ranks' stdout and every rank's stderr go to `err`.  Returns 0 when every rank exits 0; otherwise the other ranks are
    stopped (exact PIDs, terminate then kill) and the first failing rank's code is returned (124 on `timeout` seconds)."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs = []
    try:
        err_fd = err.fileno()
    except (AttributeError, OSError, ValueError):
        err_fd = None
    for r in range(nranks):
        procs.append(subprocess.Popen(list(argv), env=rank_env(r, nranks, port), stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else (err_fd if err_fd is not None else subprocess.DEVNULL),
                                      stderr=err_fd if err_fd is not None else subprocess.DEVNULL, text=(r == 0), bufsize=1 if r == 0 else -1))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    import threading

    def relay():
        for line in procs[0].stdout:  # result lines (JSON objects) to `out`; library chatter on stdout ("[Gloo] Rank 0 is connected ...") to `err`
            dst = out if line.lstrip().startswith("{") else err
            dst.write(line)
            dst.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    t0 = time.time()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 128 - bad[0]
                stop_all()
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.time() - t0 > timeout:
                rc = 124
                stop_all()
                break
            time.sleep(poll)
    except BaseException:
        stop_all()
        raise
    t.join(timeout=10.0)
    return rc
