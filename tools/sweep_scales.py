#!/usr/bin/env python3
"""The reference's own benchmark sweep (results_log.txt, produced by its test.sh: kitti_mini frames resized by 1/scale, ELAS preset
MIDDLEBURY + only-left + adaptive mean + median, disp_max = 255, full resolution "s0" and `subsampling` "s1") on this engine:
frames/s of single frames (batch 1, device memory in and out: the reference's `dmap_t`, ELAS alone) and pairs/s of streamed batches,
beside the AVG_FPS columns the reference publishes for its serial / OpenMP / CUDA builds (hardware unstated).

    python tools/sweep_scales.py > gpurun_out/scale_sweep.md          (GPU box; the real frames are tests/golden/kitti*_left.png)
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

# scale, width, height, reference AVG_FPS: CPU s0, OMP s0, CUDA s0, CPU s1, OMP s1, CUDA s1   (results_log.txt)
REF = [(0.5, 2484, 750, 1.46, 2.10, 2.50, 4.41, 5.27, 5.65), (0.6, 2070, 625, 2.20, 3.09, 3.65, 5.72, 7.49, 8.22), (0.7, 1774, 535, 3.01, 4.28, 4.99, 7.73, 10.04, 10.85),
       (0.8, 1552, 468, 3.83, 5.58, 6.34, 9.84, 12.99, 13.96), (0.9, 1380, 416, 4.88, 7.10, 8.17, 12.34, 16.37, 17.41), (1.0, 1242, 375, 5.94, 8.68, 10.03, 15.33, 20.04, 21.62),
       (1.1, 1129, 340, 7.44, 10.59, 10.99, 18.51, 24.11, 25.10), (1.2, 1035, 312, 8.38, 12.64, 15.18, 21.35, 28.26, 30.92), (1.3, 955, 288, 10.05, 14.59, 17.39, 24.88, 32.47, 35.73),
       (1.4, 887, 267, 11.61, 17.08, 20.05, 29.04, 38.45, 41.21), (1.5, 828, 250, 13.33, 22.15, 22.87, 32.83, 38.02, 46.30), (1.6, 776, 234, 16.00, 22.95, 26.17, 37.84, 49.76, 52.01),
       (1.7, 730, 220, 17.40, 25.80, 40.33, 42.68, 56.59, 58.64), (1.8, 690, 208, 19.00, 28.44, 32.83, 46.89, 62.28, 64.87), (1.9, 653, 197, 21.67, 32.34, 37.00, 53.04, 63.07, 72.58),
       (2.0, 621, 187, 23.30, 35.13, 39.01, 55.93, 75.02, 78.26), (2.1, 591, 178, 27.18, 39.60, 44.62, 64.47, 71.00, 86.62), (2.2, 564, 170, 32.51, 42.91, 48.82, 69.49, 91.38, 92.02),
       (2.3, 540, 163, 31.18, 46.32, 51.59, 74.35, 99.36, 99.85), (2.4, 517, 156, 34.73, 51.80, 57.17, 84.77, 109.59, 107.86), (2.5, 496, 150, 38.18, 56.14, 61.39, 90.04, 118.91, 117.30),
       (2.6, 477, 144, 39.91, 59.86, 67.67, 98.54, 127.25, 125.61), (2.7, 459, 138, 45.43, 66.85, 73.74, 108.56, 137.98, 136.16), (2.8, 443, 133, 48.16, 71.15, 78.78, 113.75, 151.97, 146.68),
       (2.9, 428, 129, 50.60, 66.83, 133.32, 121.91, 159.97, 155.11), (3.0, 414, 125, 56.39, 80.80, 86.89, 126.10, 163.79, 166.44)]


def resize_linear(img, w, h):
    """Bilinear, pixel centres aligned (what cv::resize INTER_LINEAR samples; the timing does not depend on the rounding)."""
    H, W = img.shape
    xs = np.clip((np.arange(w) + 0.5) * (W / w) - 0.5, 0, W - 1)
    ys = np.clip((np.arange(h) + 0.5) * (H / h) - 0.5, 0, H - 1)
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    x1, y1 = np.minimum(x0 + 1, W - 1), np.minimum(y0 + 1, H - 1)
    fx, fy = (xs - x0)[None, :], (ys - y0)[:, None]
    f = img.astype(np.float64)
    top = f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx
    bot = f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)


def main():
    import util
    pkg = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0]
    eng = importlib.import_module(pkg + ".engine")
    frames = [(util.load_png("kitti%d_left.png" % i), util.load_png("kitti%d_right.png" % i)) for i in (0, 3, 7, 10, 13, 17, 20)]  # the committed gray frames of kitti_mini
    print("| scale | image | this engine s0: frames/s (batch 1) | pairs/s (streamed) | reference s0: CPU / OMP / CUDA AVG_FPS | this engine s1: frames/s | pairs/s | reference s1: CPU / OMP / CUDA |")
    print("|---|---|---|---|---|---|---|---|")
    for scale, w, h, c0, o0, g0, c1, o1, g1 in REF:
        pairs = np.stack([np.stack([resize_linear(l, w, h), resize_linear(r, w, h)]) for l, r in frames])  # [7, 2, h, w]
        cells = []
        for sub in (0, 1):
            p = eng.SvParams.driver(255)
            p.subsampling = sub
            hm, wm = (h // 2, w // 2) if sub else (h, w)
            # single frames: the latency configuration (chunk 1), the seven frames in turn
            e = eng.StereoEngine(w, h, p, chunk=1, n_slots=2, n_streams=1)
            L = [torch.from_numpy(np.ascontiguousarray(q[0][None])).cuda() for q in pairs]
            R = [torch.from_numpy(np.ascontiguousarray(q[1][None])).cuda() for q in pairs]
            d1 = torch.empty((1, hm, wm), dtype=torch.float32, device="cuda")
            d2 = torch.empty_like(d1)
            for i in range(7):
                e.process_device(L[i], R[i], d1, d2)
            ts = []
            for rep in range(6):
                for i in range(7):
                    torch.cuda.synchronize()
                    a = time.perf_counter()
                    e.process_device(L[i], R[i], d1, d2)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - a)
            e.close()
            fps = float(np.mean(1.0 / np.array(ts)))  # the reference's AVG_FPS: mean over frames of 1 / t
            # streamed batches of 128 pairs (the seven frames in turn)
            B = 128
            bl = torch.from_numpy(np.ascontiguousarray(np.concatenate([pairs[:, 0]] * 19)[:B])).cuda()
            br = torch.from_numpy(np.ascontiguousarray(np.concatenate([pairs[:, 1]] * 19)[:B])).cuda()
            o1_, o2_ = torch.empty((B, hm, wm), dtype=torch.float32, device="cuda"), torch.empty((B, hm, wm), dtype=torch.float32, device="cuda")
            e = eng.StereoEngine(w, h, p)
            e.process_device(bl, br, o1_, o2_)
            torch.cuda.synchronize()
            n, a = 0, time.perf_counter()
            while time.perf_counter() - a < 1.5:
                for _ in range(8):
                    e.submit_device(bl, br, o1_, o2_)
                e.wait()
                n += 8
            torch.cuda.synchronize()
            rate = n * B / (time.perf_counter() - a)
            e.close()
            cells.append((fps, rate))
        print("| %.1f | %dx%d | %.0f | %.0f | %.2f / %.2f / %.2f | %.0f | %.0f | %.2f / %.2f / %.2f |" % (scale, w, h, cells[0][0], cells[0][1], c0, o0, g0, cells[1][0], cells[1][1], c1, o1, g1), flush=True)


if __name__ == "__main__":
    main()
