import ctypes, glob, importlib, mmap, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
PKG = bench.PKG
eng = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth")
libc = ctypes.CDLL("libc.so.6", use_errno=True)
def node_of(addr):
    pages = (ctypes.c_void_p * 1)(addr & ~4095)
    status = (ctypes.c_int * 1)(-99)
    rc = libc.syscall(279, 0, 1, pages, None, status, 0)  # move_pages(query)
    return status[0] if rc == 0 else "err%d" % ctypes.get_errno()
print("cpus allowed", len(os.sched_getaffinity(0)), "cpu now", libc.sched_getcpu())
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(n.split("/")[-1], open(n + "/cpulist").read().strip())
props = torch.cuda.get_device_properties(0)
bdf = None
try:
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    hip.hipDeviceGetPCIBusId(buf, 64, 0)
    bdf = buf.value.decode().lower()
    print("gpu", bdf, "numa_node", open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
except Exception as ex:
    print("pci lookup failed", ex)
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception: pass
W, H, D, B = 1242, 375, 128, 256
batch = synth.make_batch(1000, 32, H, W, D); batch = np.concatenate([batch] * 8)
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1))
for kind in ("pinned", "pageable", "pinned"):
    alloc = eng.pinned_array if kind == "pinned" else (lambda shape, dt: np.zeros(shape, dt))
    L, R = alloc((B, H, W), np.uint8), alloc((B, H, W), np.uint8)
    L[:], R[:] = batch[:, 0], batch[:, 1]
    d1 = alloc((B, H, W), np.float32); d1[:] = 0
    print(kind, "nodes: L", node_of(L.ctypes.data), "R", node_of(R.ctypes.data), "d1", node_of(d1.ctypes.data), node_of(d1.ctypes.data + d1.nbytes - 1), "cpu now", libc.sched_getcpu())
    e.process_host(L, R, want_d2=False, d1=d1)
    rates = []
    for rep in range(6):
        t0 = time.perf_counter()
        for _ in range(8):
            e.submit_host(L, R, d1, None)
        e.wait()
        rates.append(B * 8 / (time.perf_counter() - t0))
    print(kind, "d1 rates", [int(r) for r in rates], flush=True)
e.close()
# raw DMA rate to page-locked memory on each NUMA node (mmap + mbind + hipHostRegister)
n = 256 << 20
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
nodes = [int(p.split("node")[-1]) for p in glob.glob("/sys/devices/system/node/node*")]
for node in sorted(nodes):
    mm = mmap.mmap(-1, n)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    mask = (ctypes.c_ulong * 16)()
    mask[node // 64] = 1 << (node % 64)
    rc = libc.syscall(237, ctypes.c_void_p(addr), ctypes.c_ulong(n), 2, mask, 1024, 0)  # mbind(MPOL_BIND)
    arr = np.frombuffer(mm, dtype=np.uint8); arr[:] = 1
    got = node_of(addr)
    rr = hip.hipHostRegister(ctypes.c_void_p(addr), ctypes.c_size_t(n), 0)
    t = torch.from_numpy(arr)
    s = torch.cuda.Stream()
    res = {}
    for name, fn in (("d2h", lambda: hip.hipMemcpyAsync(ctypes.c_void_p(addr), ctypes.c_void_p(dev.data_ptr()), ctypes.c_size_t(n), 2, ctypes.c_void_p(s.cuda_stream))),
                     ("h2d", lambda: hip.hipMemcpyAsync(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(n), 1, ctypes.c_void_p(s.cuda_stream)))):
        fn(); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): fn()
        s.synchronize()
        res[name] = round(4 * n / (time.perf_counter() - t0) / 1e9, 1)
    print("node", node, "mbind rc", rc, "landed on", got, "register rc", rr, res, flush=True)
    hip.hipHostUnregister(ctypes.c_void_p(addr))
    del t, arr
