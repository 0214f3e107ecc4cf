"""Starts one fresh process per GPU (rank) for a multi-GPU run that was invoked as a plain `python bench.py --gpus N`.

Standard library only, and it never touches the GPU: a process that has initialised HIP must not be re-executed (on this
pool that takes the machine down), so the parent stays a pure launcher - it spawns N children with the environment
torch.distributed's env:// rendezvous reads (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
relays rank 0's JSON line (its other stdout lines go to stderr) and returns the first non-zero exit code after stopping the other ranks.
`python -m torch.distributed.run ... bench.py` keeps working: then WORLD_SIZE is already set and no launcher runs.

The reference has no counterpart (one GPU, blocking copies: src/parallel_includes/elas/elas_gpu.cu:537-563).
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
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

    def relay():
        for line in procs[0].stdout:  # result lines (JSON objects) to `out`; library chatter on stdout ("[Gloo] Rank 0 is connected ...") to `err`
            dst = out if line.lstrip().startswith("{") else err
            dst.write(line)
            dst.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    t0 = time.time()
    rc = 0
    # A SIGTERM / SIGHUP to the launcher (a harness timeout, a closed terminal) must not leave N GPU-holding ranks behind: the
    # handler turns the signal into an exception in the loop below, which stops the children (exact PIDs) and exits 128 + signal.
    restore = {}

    def on_signal(signum, _frame):
        raise SystemExit(128 + signum)

    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGHUP):
            restore[sg] = signal.signal(sg, on_signal)
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
    finally:
        for sg, old in restore.items():
            signal.signal(sg, old)
    t.join(timeout=10.0)
    return rc


def usable_cpus():
    """CPUs this process may use: the cgroup CPU quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError, IndexError):
        pass
    return max(1, n)


def gpu_numa_cpus(local_rank=0):
    """CPUs of the NUMA node the local_rank-th usable GPU hangs off, read from sysfs WITHOUT touching the GPU: KFD topology nodes in
    order (the order the HIP runtime enumerates), those whose render node this process may open.  Empty set when anything is missing."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, n, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) == 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            if minor < 0 or not os.access("/dev/dri/renderD%d" % minor, os.R_OK | os.W_OK):
                continue
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            gpus.append("%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % gpus[local_rank]).read())
        if node < 0:
            return set()
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus
    except (OSError, ValueError, IndexError, KeyError):
        pass
    try:  # second source: the DRM render nodes this process may open, in minor order
        minors = sorted(int(n[7:]) for n in os.listdir("/sys/class/drm") if n.startswith("renderD") and os.access("/dev/dri/" + n, os.R_OK | os.W_OK))
        node = int(open("/sys/class/drm/renderD%d/device/numa_node" % minors[local_rank]).read())
        if node < 0:
            return set()
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus
    except (OSError, ValueError, IndexError):
        return set()


def restrict_to_host_share(share, local_rank=0):
    """Gives this process the host budget one rank has on a node with `share` ranks: CPU affinity = usable_cpus() / share CPUs (of
    the GPU's NUMA node when sysfs tells which, else of the mask; slice local_rank + 1 of them) and LOCAL_WORLD_SIZE = share.  To be called BEFORE
    anything touches the GPU or starts a thread pool.  Returns the CPUs kept."""
    allowed = sorted(os.sched_getaffinity(0))
    n = max(1, usable_cpus() // max(1, share))
    near = [c for c in allowed if c in gpu_numa_cpus(local_rank)]
    cpus = near if len(near) >= n else allowed
    # rank r takes slice r + 1 (mod share) of the list when the list holds `share` slices: the ranks of a node get distinct slices, and a single
    # sampled rank (bench.py's value_at_host_share_N) does not sit on CPU 0 with the machine's interrupts and housekeeping (one measurement
    # moved between 42 600 and 45 900 pairs/s with it)
    first = ((local_rank + 1) % max(1, share)) * n if len(cpus) >= n * max(1, share) else 0
    keep = cpus[first:first + n]
    os.sched_setaffinity(0, keep)
    os.environ["LOCAL_WORLD_SIZE"] = str(share)
    return keep


def run_host_share_child(argv, timeout=900):
    """Runs `argv` (a bench.py command that restricts itself with restrict_to_host_share) as ONE fresh child and returns its JSON line
    as a dict (None when it failed).  The caller must not have initialised the GPU yet: the child has the device to itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        p = subprocess.run(list(argv), env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None
    for line in reversed(p.stdout.splitlines()):
        if line.lstrip().startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                break
    sys.stderr.write("host-share child failed (exit %d): %s\n" % (p.returncode, p.stderr[-2000:]))
    return None
