"""Seeded synthetic KITTI-shaped stereo pairs (SURVEY.md §8d "Synthetic pair generator").

There is no network for datasets, so every benchmark and most parity tests run on pairs made here.
The generator is pure numpy (PCG64 via ``default_rng(seed)``) and therefore reproducible on the GPU box.

Scene: a blurred random texture (so that every lattice point has gradient energy), a slanted ground
plane below 0.45*H, a far background above it and K fronto-parallel rectangles painted by increasing
disparity.  The disparity field is defined in the *right* image frame and the right image is the left
texture sampled at ``u + d`` (so a left pixel u matches right pixel u - d), plus rounded Gaussian noise.
"""
import numpy as np


def _box3(a):
    """3x3 integer box mean (floor), edge-replicated."""
    p = np.pad(a.astype(np.int32), 1, mode="edge")
    s = sum(p[dy:dy + a.shape[0], dx:dx + a.shape[1]] for dy in range(3) for dx in range(3))
    return (s // 9).astype(np.uint8)


def disparity_field(rng, H, W, D, n_rect=12, scale=1):
    v = np.arange(H, dtype=np.float64)[:, None]
    ground = np.clip(np.round(0.35 * (v - 0.45 * H) * D / (0.55 * H)), 2, D - 8)
    d = np.where(v > 0.45 * H, ground, 2.0) * np.ones((1, W))
    rects = []
    for _ in range(n_rect):
        w = int(rng.integers(40, 201)) * scale
        h = int(rng.integers(30, 121)) * scale
        x0 = int(rng.integers(0, max(1, W - w)))
        y0 = int(rng.integers(0, max(1, H - h)))
        dd = int(rng.integers(8, max(9, D - 8) + 1))
        rects.append((dd, x0, y0, w, h))
    for dd, x0, y0, w, h in sorted(rects):
        d[y0:y0 + h, x0:x0 + w] = dd
    return d.astype(np.int32)


def make_pair(seed, H=375, W=1242, D=128, noise=1.5, scale=1):
    """Returns (left, right) uint8 [H, W] gray images with true disparities in [2, D-8]."""
    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 256, (H, W + D), dtype=np.uint8)
    tex = _box3(_box3(tex))
    left = np.ascontiguousarray(tex[:, :W])
    d = disparity_field(rng, H, W, D, scale=scale)
    cols = np.arange(W, dtype=np.int64)[None, :] + d
    right = np.take_along_axis(tex, cols, axis=1).astype(np.int32)
    if noise > 0:
        right = right + np.rint(rng.normal(0.0, noise, (H, W))).astype(np.int32)
    right = np.clip(right, 0, 255).astype(np.uint8)
    return left, right


def make_batch(seed0, B, H=375, W=1242, D=128, scale=1):
    """[B, 2, H, W] uint8: pair i uses seed seed0 + i (KITTI-shaped: 1000+i; 4K: 5000+i with scale=3)."""
    out = np.empty((B, 2, H, W), np.uint8)
    for i in range(B):
        out[i, 0], out[i, 1] = make_pair(seed0 + i, H, W, D, scale=scale)
    return out
