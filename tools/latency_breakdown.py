import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"
eng = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth")
W, H, D = 1242, 375, 128
b = synth.make_batch(1000, 4, H, W, D)
if os.environ.get("LAT_KITTI"):  # the committed kitti_mini pair 0 instead of synthetic pairs
    from PIL import Image
    g = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests', 'golden')
    gl, gr = np.asarray(Image.open(g + '/kitti0_left.png')), np.asarray(Image.open(g + '/kitti0_right.png'))
    b = np.stack([np.stack([gl, gr])] * 4)
left = torch.from_numpy(np.ascontiguousarray(b[:, 0])).cuda(); right = torch.from_numpy(np.ascontiguousarray(b[:, 1])).cuda()
e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=int(os.environ.get("LAT_WORKERS", "4")), chunk=1, n_streams=1, n_slots=2)
d1 = torch.zeros((1, H, W), dtype=torch.float32, device='cuda'); d2 = torch.zeros_like(d1)
for i in range(20): e.process_device(left[i % 4:i % 4 + 1], right[i % 4:i % 4 + 1], d1, d2)
ts = []
for i in range(200):
    t0 = time.perf_counter(); e.process_device(left[i % 4:i % 4 + 1], right[i % 4:i % 4 + 1], d1, d2); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("latency ms median %.3f p10 %.3f p99 %.3f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 99)))
e.timing(True)
for i in range(50): e.process_device(left[i % 4:i % 4 + 1], right[i % 4:i % 4 + 1], d1, d2)
kt = e.kernel_times()
tot = 0
for k, (ms, calls) in sorted(kt.items(), key=lambda x: -x[1][0]):
    if calls:
        print("%-28s %7.1f us/call  calls %d" % (k, 1e3 * ms / calls, calls)); 
        if not k.startswith('host'): tot += 1e3 * ms / 50
print("sum of kernel time per frame: %.1f us" % tot)
e.close()
