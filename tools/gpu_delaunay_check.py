import os, sys, importlib, numpy as np, time
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import util
eng = util.pkg("engine")
import torch
torch.cuda.init()
rng = np.random.default_rng(41)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    n = int(rng.integers(3, int(os.environ.get("GDL_MAXN", "3900")) if it % 10 == 0 else 500))
    k = it % 4
    if k == 0: pts = rng.integers(0, 250, (n, 2)) * 5
    elif k == 1: pts = np.stack([rng.integers(-50, 1300, n), rng.integers(0, 75, n) * 5], 1)
    elif k == 2: pts = np.stack([rng.integers(0, 12, n) * 5, rng.integers(0, 12, n) * 5], 1)
    else: pts = rng.integers(0, 3000, (n, 2))
    if len(np.unique(pts, axis=0)) < 3: continue
    a = eng.host_delaunay(pts)
    b, ms = eng.gpu_delaunay(pts)
    if a.shape != b.shape or not np.array_equal(a, b):
        bad += 1
        print("MISMATCH", it, n, a.shape, b.shape, flush=True)
        if bad > 5: break
print("gpu delaunay vs host: bad", bad)
g = util.golden_npz("kitti0_d128")["support"].reshape(-1, 3)
for side in (0, 1):
    pts = np.stack([g[:, 0] - (g[:, 2] if side else 0), g[:, 1]], 1)
    a = eng.host_delaunay(pts)
    b, ms1 = eng.gpu_delaunay(pts, reps=1)
    b2, ms128 = eng.gpu_delaunay(pts, reps=128)
    print("kitti0 side", side, "equal", np.array_equal(a, b), np.array_equal(a, b2), "kernel ms: 1 set %.3f, 128 sets %.3f" % (ms1, ms128))
