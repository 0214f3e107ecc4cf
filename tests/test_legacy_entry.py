"""The reference's Python entry point / exported symbols on top of the HIP library: mirrors the
reference's own smoke test (tests/test_demo.py) but with assertions."""
import ctypes
import os

import numpy as np
import pytest

import util
from pyoracle import ElasParams


def _q(eng, w, h, scale=1.0, variant=1):
    eng.share_hip_runtime_with_torch()
    L = ctypes.CDLL(eng.LIB_PATH)
    L.sv_debug_stereo_rectify.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    Q = np.zeros(16)
    P = np.zeros(24)
    yml = os.path.join(os.path.dirname(eng.LIB_PATH), "stereo_vision", "data", "kitti_2011_09_26.yml").encode()
    assert L.sv_debug_stereo_rectify(yml, w, h, scale, variant, Q.ctypes.data, P.ctypes.data) == 0
    return Q.reshape(4, 4), P[:12].reshape(3, 4), P[12:].reshape(3, 4)


def test_stereo_rectify_structure():
    """Properties every stereoRectify result has (no OpenCV here to compare numbers with: parity unpinned)."""
    eng = util.pkg("engine")
    util.pkg("build").build()
    Q, P1, P2 = _q(eng, 1242, 375)
    f, cx, cy = Q[2, 3], -Q[0, 3], -Q[1, 3]
    assert Q[0, 0] == 1 and Q[1, 1] == 1 and Q[2, 2] == 0 and Q[3, 3] == 0  # CALIB_ZERO_DISPARITY: cx1 == cx2
    assert abs(1.0 / Q[3, 2] - 0.5372) < 1e-3                               # 1/Tx = |T| of the rig (0.537 m baseline)
    assert P1[0, 0] == f and P1[0, 2] == cx and P1[1, 2] == cy and P2[0, 2] == cx
    assert abs(P2[0, 3] / f + 1.0 / Q[3, 2]) < 1e-9                          # P2[0][3] = Tx * f
    assert 600 < f < 1400 and 0 < cx < 1242 and 0 < cy < 375
    Qh, _, _ = _q(eng, 621, 187, scale=2.0)
    assert abs(Qh[2, 3] / f - 0.5) < 0.01                                    # half-size images: half the focal length


def _rodrigues(v):
    """Rotation vector -> matrix (Rodrigues' formula), numpy."""
    th = float(np.linalg.norm(v))
    if th < np.finfo(float).eps:
        return np.eye(3)
    k = np.asarray(v, float) / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def _rodrigues_inv(R):
    """Matrix -> rotation vector as cvRodrigues2 does it: nearest rotation by SVD first, then axis * angle."""
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = np.linalg.norm(r) / 2, np.clip((np.trace(R) - 1) / 2, -1, 1)
    return r * (np.arccos(c) / (2 * s))


def _undistort(K, D, pts, R=None, P=None, iters=5):
    """cv::undistortPoints, 5-coefficient model: fixed-point iteration, optional R and P[:, :3]; float32 result like its CV_32FC2 destination."""
    x0 = (pts[:, 0].astype(np.float64) - K[0, 2]) / K[0, 0]
    y0 = (pts[:, 1].astype(np.float64) - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    k1, k2, p1, p2, k3 = D
    for _ in range(iters):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    M = np.eye(3) if R is None else np.asarray(R, float)
    if P is not None:
        M = np.asarray(P, float)[:, :3] @ M
    h = M @ np.stack([x, y, np.ones_like(x)])
    return np.stack([h[0] / h[2], h[1] / h[2]], 1).astype(np.float32)


def _stereo_rectify_numpy(K1, D1, K2, D2, R, T, w, h):
    """cv::stereoRectify(..., CALIB_ZERO_DISPARITY, alpha = 0, newImageSize = imageSize) for a horizontal rig, OpenCV >= 3.4.2 rules,
    derived here independently of csrc/calib.cpp (SVD instead of its Newton polar factor, vectorised corners and rectangles)."""
    r_r = _rodrigues(-0.5 * _rodrigues_inv(R))
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1
    uu = np.zeros(3)
    uu[idx] = 1.0 if t[idx] > 0 else -1.0
    ww = np.cross(t, uu)
    nw = np.linalg.norm(ww)
    if nw > 0:
        ww *= np.arccos(abs(t[idx]) / np.linalg.norm(t)) / nw
    wR = _rodrigues(ww)
    R1, R2 = wR @ r_r.T, wR @ r_r
    t = R2 @ T
    fc = (K1[idx ^ 1, idx ^ 1] + K2[idx ^ 1, idx ^ 1]) * 0.5  # new size == old size: ratio 1/2 of the sum
    corners = np.array([[0, 0], [w, 0], [0, h], [w, h]], np.float32)
    cc = []
    for K, D, Rk in ((K1, D1, R1), (K2, D2, R2)):
        u = _undistort(K, D, corners).astype(np.float64)
        X = Rk @ np.stack([u[:, 0], u[:, 1], np.ones(4)])
        proj = np.stack([(fc * X[0] / X[2]).astype(np.float32), (fc * X[1] / X[2]).astype(np.float32)], 1).astype(np.float64)
        cc.append(np.array([w / 2.0 - proj[:, 0].sum() / 4, h / 2.0 - proj[:, 1].sum() / 4]))
    c0 = (cc[0] + cc[1]) * 0.5  # CALIB_ZERO_DISPARITY
    P1 = np.array([[fc, 0, c0[0], 0], [0, fc, c0[1], 0], [0, 0, 1, 0]])
    P2 = P1.copy()
    P2[idx, 3] = t[idx] * fc
    g = np.arange(9, dtype=np.float32)
    grid = np.stack([np.tile(g * np.float32(w) / np.float32(8), 9), np.repeat(g * np.float32(h) / np.float32(8), 9)], 1)
    s0 = 0.0
    for K, D, Rk, Pk in ((K1, D1, R1, P1), (K2, D2, R2, P2)):
        q = _undistort(K, D, grid, Rk, Pk).reshape(9, 9, 2)  # [y][x]
        ix0, ix1 = q[:, 0, 0].max(), q[:, 8, 0].min()
        iy0, iy1 = q[0, :, 1].max(), q[8, :, 1].min()
        s0 = max(s0, c0[0] / (c0[0] - ix0), c0[1] / (c0[1] - iy0), (w - c0[0]) / (ix1 - c0[0]), (h - c0[1]) / (iy1 - c0[1]))
    f = fc * s0  # alpha = 0: zoom until only valid pixels remain
    P1[0, 0] = P1[1, 1] = P2[0, 0] = P2[1, 1] = f
    P2[idx, 3] *= s0
    Q = np.array([[1, 0, 0, -c0[0]], [0, 1, 0, -c0[1]], [0, 0, 0, f], [0, 0, -1.0 / t[idx], 0.0]])
    return Q, P1, P2, t


def _read_calibration(path):
    import re
    txt = open(path).read()

    def arr(name, n):
        m = re.search(name + r":[^\[]*\[([^\]]*)\]", txt)
        return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",")], float)[:n]

    return (arr("K1", 9).reshape(3, 3), arr("D1", 5), arr("K2", 9).reshape(3, 3), arr("D2", 5), arr("R", 9).reshape(3, 3), arr("T", 3))


def test_stereo_rectify_numbers_for_the_bundled_calibration():
    """Q, P1, P2 of csrc/calib.cpp for the KITTI calibration against the same published algorithm derived a second time in numpy,
    and against what has a closed form: 1 / Q[3][2] = |T| exactly (the rectifying rotations put the baseline on the x axis), the focal
    length before the alpha = 0 zoom = (fy1 + fy2) / 2 (OpenCV >= 3.4.2), P2[0][3] = -|T| f.  (OpenCV itself is not installed: the
    arithmetic stays unpinned against the library, but a slip in the C++ restatement now shows.)"""
    eng = util.pkg("engine")
    util.pkg("build").build()
    yml = os.path.join(os.path.dirname(eng.LIB_PATH), "stereo_vision", "data", "kitti_2011_09_26.yml")
    K1, D1, K2, D2, R, T = _read_calibration(yml)
    for (w, h, scale) in ((1242, 375, 1.0), (621, 187, 2.0)):
        Q, P1, P2 = _q(eng, w, h, scale=scale)
        k1, k2 = K1.copy(), K2.copy()
        k1[:2] /= scale
        k2[:2] /= scale
        Qn, P1n, P2n, t = _stereo_rectify_numpy(k1, D1, k2, D2, R, T, w, h)
        assert abs(1.0 / Q[3, 2] - np.linalg.norm(T)) < 1e-12 and abs(t[0] + np.linalg.norm(T)) < 1e-12 and abs(t[1]) < 1e-12 and abs(t[2]) < 1e-12
        assert np.allclose(Q, Qn, rtol=1e-9, atol=1e-9), (Q, Qn)
        assert np.allclose(P1, P1n, rtol=1e-9, atol=1e-9) and np.allclose(P2, P2n, rtol=1e-9, atol=1e-9)
        assert abs(P2[0, 3] + np.linalg.norm(T) * Q[2, 3]) < 1e-9
        assert Q[2, 3] >= (k1[1, 1] + k2[1, 1]) / 2  # the zoom only ever magnifies at alpha = 0


@pytest.mark.gpu
def test_generate_point_cloud_like_reference_smoke_test():
    eng = util.pkg("engine")
    svmod = util.pkg("stereo_vision")
    L, R = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
    H, W = L.shape
    s = svmod.stereo_vision(objectTracking=False, width=W, height=H)
    try:
        pts = s.generatePointCloud(np.repeat(L[:, :, None], 3, 2), np.repeat(R[:, :, None], 3, 2))
        assert pts.shape == (W * H, 3) and pts.dtype == np.float64
        pts = np.array(pts)
        dmap = s.last_disparity_u8()
        s.sv.sv_legacy_Q.restype = ctypes.POINTER(ctypes.c_double)
        Q = np.ctypeslib.as_array(s.sv.sv_legacy_Q(), shape=(16,)).reshape(4, 4).copy()
        means = s.object_positions([(600, 200, 24, 16), (-5, 360, 30, 40), (1230, 0, 50, 9)])
        pts2 = np.array(s.generatePointCloud(np.repeat(L[:, :, None], 3, 2), np.repeat(R[:, :, None], 3, 2)))
    finally:
        s.close()
    assert np.array_equal(pts, pts2, equal_nan=True)
    # gray(B=G=R=g) == g with the 15-bit weights (they sum to 32768), so the disparity equals the golden kitti0 result at disp_max 255
    final = util.golden_npz("kitti0_d256")["final1"].reshape(H, W)
    want = np.clip(np.rint(final * np.float32(4.0)), 0, 255).astype(np.uint8)
    assert np.array_equal(dmap, want)
    jj, ii = np.mgrid[0:H, 0:W]
    V = np.stack([ii.ravel().astype(np.float64), jj.ravel().astype(np.float64), want.ravel().astype(np.float64), np.ones(W * H)], 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        pos = ((Q[:, 0:1] * V[0] + Q[:, 1:2] * V[1]) + Q[:, 2:3] * V[2]) + Q[:, 3:4]
        exp = (pos[:3] / pos[3]).T
    assert np.array_equal(pts, exp, equal_nan=True)
    assert np.isfinite(pts[want.ravel() > 0]).all()
    # object positions (stereo_vision.cpp:261-278): sequential double sums, columns outer / rows inner, over clamped boxes
    boxes = [(600, 200, 24, 16), (-5, 360, 30, 40), (1230, 0, 50, 9)]
    P3 = pts.reshape(H, W, 3)
    for (x, y, w, h), got in zip(boxes, means):
        i0, i1 = min(max(x, 0), W - 1), min(max(x + w, 0), W - 1)
        j0, j1 = min(max(y, 0), H - 1), min(max(y + h, 0), H - 1)
        acc = [0.0, 0.0, 0.0]
        with np.errstate(all="ignore"):
            for i in range(i0, i1):
                for j in range(j0, j1):
                    for c in range(3):
                        acc[c] = float(np.float64(acc[c]) + P3[j, i, c])
            exp_m = np.array(acc) / np.float64((i1 - i0) * (j1 - j0))
        assert np.array_equal(got, exp_m, equal_nan=True)


@pytest.mark.gpu
def test_reference_smoke_input_zero_images():
    """tests/test_demo.py of the reference: all-zero images must go through (there: (1242,375) 2-D arrays, which its
    own cvtColor would reject; here proper (375,1242,3) images)."""
    svmod = util.pkg("stereo_vision")
    s = svmod.stereo_vision(objectTracking=False, width=1242, height=375)
    try:
        z = np.zeros((375, 1242, 3), np.uint8)
        pts = np.array(s.generatePointCloud(z, z))
        dmap = s.last_disparity_u8()
    finally:
        s.close()
    assert pts.shape == (1242 * 375, 3) and not dmap.any()


@pytest.mark.gpu
def test_generate_point_cloud_subsampling(oracle):
    """subsampling=True (sv.py constructor flag -> Elas::parameters::subsampling, stereo_vision.cpp:309): Elas fills the
    first (W/2)*(H/2) floats of the driver's zeroed full-size leftdpf (:304, elas.h:160-161) and the driver converts the
    whole buffer as it stands (:316) - reproduce exactly that buffer."""
    svmod = util.pkg("stereo_vision")
    L, R = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
    H, W = L.shape
    s = svmod.stereo_vision(objectTracking=False, width=W, height=H, subsampling=True)
    try:
        pts = np.array(s.generatePointCloud(np.repeat(L[:, :, None], 3, 2), np.repeat(R[:, :, None], 3, 2)))
        dmap = s.last_disparity_u8()
    finally:
        s.close()
    p = ElasParams.driver(255)
    p.subsampling = 1
    d1, _, _ = oracle.process(p, L, R)
    assert d1.shape == (H // 2, W // 2)
    buf = np.zeros(W * H, np.float32)
    buf[:d1.size] = d1.ravel()
    want = np.clip(np.rint(buf * np.float32(4.0)), 0, 255).astype(np.uint8).reshape(H, W)
    assert np.array_equal(dmap, want)
    assert pts.shape == (W * H, 3)


@pytest.mark.gpu
def test_reproject_batch_device():
    """Batched conversion + reprojection against the same arithmetic in numpy doubles (operation order of stereo_vision.cpp:233-256
    and of projectParallel, stereo_vision.cu:200-211)."""
    import torch

    eng = util.pkg("engine")
    rng = np.random.default_rng(4)
    B, H, W = 3, 37, 101
    d = rng.uniform(-12, 70, (B, H, W)).astype(np.float32)
    d[0, :5] = -10
    d[1, 3, 4] = 63.875  # x4 = 255.5 -> rounds to even 256 -> saturates
    d[1, 3, 5] = 0.125   # x4 = 0.5 -> rounds to 0
    d[1, 3, 6] = 0.375   # x4 = 1.5 -> rounds to 2
    Q = np.array([[1, 0, 0, -50.5], [0, 1, 0, -18.25], [0, 0, 0, 721.5], [0, 0, 1.86, 0.01]])
    XR = np.array([[0.1, -0.99, 0.02], [0.03, 0.2, -0.97], [0.99, 0.05, 0.11]])
    XT = np.array([0.27, -0.08, 1.65])
    want_u8 = np.clip(np.rint(d * np.float32(4.0)), 0, 255).astype(np.uint8)
    jj, ii = np.mgrid[0:H, 0:W]
    x, y, v = ii.astype(np.float64)[None], jj.astype(np.float64)[None], want_u8.astype(np.float64)
    pos = [((Q[r, 0] * x + Q[r, 1] * y) + Q[r, 2] * v) + Q[r, 3] for r in range(4)]
    with np.errstate(divide="ignore", invalid="ignore"):
        X, Y, Z = pos[0] / pos[3], pos[1] / pos[3], pos[2] / pos[3]
        plain = np.stack([X, Y, Z], -1)
        xf = np.stack([((XR[r, 0] * X + XR[r, 1] * Y) + XR[r, 2] * Z) + XT[r] for r in range(3)], -1)
    dmap, pts = eng.reproject(torch.from_numpy(d).cuda(), Q)
    assert np.array_equal(dmap.cpu().numpy(), want_u8)
    assert np.array_equal(pts.cpu().numpy(), plain, equal_nan=True)
    _, pts2 = eng.reproject(torch.from_numpy(d).cuda(), Q, XR, XT, want_dmap=False)
    assert np.array_equal(pts2.cpu().numpy(), xf, equal_nan=True)


def _gray_cv4(rgb):
    a = rgb.astype(np.int64)  # OpenCV 4.x BGR2GRAY, 15-bit fixed point (tests/golden/make_golden.py:gray_cv4)
    return ((a[..., 2] * 3735 + a[..., 1] * 19235 + a[..., 0] * 9798 + 16384) >> 15).astype(np.uint8)


def test_colour_crop_fixture_matches_gray_fixture():
    """The committed colour crop of kitti_mini pair 0 and the committed gray pair are the same pixels: gray_cv4(colour) == gray."""
    for side in ("left", "right"):
        rgb = util.load_png("kitti0_crop_color_%s.png" % side)
        assert rgb.shape == (128, 320, 3)
        assert (rgb[..., 0] != rgb[..., 2]).mean() > 0.9  # R != B almost everywhere: a swapped channel order cannot pass
        assert np.array_equal(_gray_cv4(rgb), util.load_png("kitti0_%s.png" % side)[150:278, 400:720])


@pytest.mark.gpu
def test_generate_point_cloud_colour_input(oracle):
    """Distinct B, G, R through k_bgra_to_gray (stereo_vision.cpp:338-339): the u8 disparity image equals the oracle's result on
    gray_cv4 of the same colour images (disp_max 255 as the driver runs it) - a B/R swap or wrong weights change the gray
    image and with it the map."""
    svmod = util.pkg("stereo_vision")
    rgb_l, rgb_r = util.load_png("kitti0_crop_color_left.png"), util.load_png("kitti0_crop_color_right.png")
    H, W = rgb_l.shape[:2]
    gl, gr = _gray_cv4(rgb_l), _gray_cv4(rgb_r)
    swapped = _gray_cv4(rgb_l[..., ::-1])
    assert (swapped != gl).mean() > 0.5  # the test has teeth
    s = svmod.stereo_vision(objectTracking=False, width=W, height=H)
    try:
        pts = s.generatePointCloud(rgb_l[..., ::-1], rgb_r[..., ::-1])  # BGR, as cv2.imread hands images to the reference's wrapper
        assert pts.shape == (W * H, 3)
        dmap = s.last_disparity_u8()
    finally:
        s.close()
    d1, _, _ = oracle.process(ElasParams.driver(255), gl, gr)
    want = np.clip(np.rint(d1 * np.float32(4.0)), 0, 255).astype(np.uint8)
    assert (want > 0).mean() > 0.3
    assert np.array_equal(dmap, want)


def _lib(eng):
    eng.share_hip_runtime_with_torch()
    return ctypes.CDLL(eng.LIB_PATH)


def test_undistort_rectify_map_against_formula():
    """initUndistortRectifyMap restatement (stereo_vision.cpp:477-478): identity camera -> identity map; with distortion and a
    rotation -> the documented pinhole + (k1,k2,p1,p2,k3) formula evaluated directly in numpy (no OpenCV here: parity unpinned)."""
    eng = util.pkg("engine")
    util.pkg("build").build()
    L = _lib(eng)
    L.sv_debug_undistort_map.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    w, h = 97, 53
    K = np.array([[420.0, 0, 48.5], [0, 415.0, 26.0], [0, 0, 1]])
    P = np.hstack([K, np.zeros((3, 1))])
    mx, my = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    assert L.sv_debug_undistort_map(K.ctypes.data, np.zeros(5).ctypes.data, np.eye(3).ctypes.data, P.ctypes.data, w, h, mx.ctypes.data, my.ctypes.data) == 0
    jj, ii = np.meshgrid(np.arange(w), np.arange(h))
    assert np.abs(mx - jj).max() < 1e-3 and np.abs(my - ii).max() < 1e-3
    D = np.array([-0.28, 0.09, 1e-3, -5e-4, -0.01])
    a = 0.02
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    P2 = np.array([[400.0, 0, 50.0, -30.0], [0, 400.0, 25.0, 0], [0, 0, 1, 0]])
    assert L.sv_debug_undistort_map(K.ctypes.data, D.ctypes.data, R.ctypes.data, P2.ctypes.data, w, h, mx.ctypes.data, my.ctypes.data) == 0
    iR = np.linalg.inv(P2[:, :3] @ R)
    xyz = iR @ np.stack([jj.ravel(), ii.ravel(), np.ones(w * h)]).astype(np.float64)
    x, y = xyz[0] / xyz[2], xyz[1] / xyz[2]
    r2 = x * x + y * y
    kr = 1 + ((D[4] * r2 + D[1]) * r2 + D[0]) * r2
    xd = x * kr + D[2] * 2 * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * kr + D[2] * (r2 + 2 * y * y) + D[3] * 2 * x * y
    assert np.abs(mx.ravel() - (K[0, 0] * xd + K[0, 2])).max() < 2e-3 and np.abs(my.ravel() - (K[1, 1] * yd + K[1, 2])).max() < 2e-3


def _remap_linear_u8(src, mapx, mapy):
    """cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) for CV_8UC1 / CV_32FC1 maps: 5 fractional bits, weights sum to 2^15."""
    H, W = src.shape
    sx, sy = np.rint(mapx * np.float32(32)).astype(np.int64), np.rint(mapy * np.float32(32)).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    pad = np.zeros((H + 2, W + 2), np.int64)

    def at(x, y):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        return np.where(ok, src[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)].astype(np.int64), 0)

    v = at(ix, iy) * (32 - fx) * (32 - fy) * 32 + at(ix + 1, iy) * fx * (32 - fy) * 32 + at(ix, iy + 1) * (32 - fx) * fy * 32 + at(ix + 1, iy + 1) * fx * fy * 32
    del pad
    return ((v + (1 << 14)) >> 15).astype(np.uint8)


@pytest.mark.gpu
def test_generate_point_cloud_with_rectification(oracle):
    """sv_legacy_set_rectify(1): the remap the reference has commented out (stereo_vision.cpp:341) in front of the matcher.  The
    gray images the matcher received equal the fixed-point bilinear remap of gray_cv4(colour) through the library's own maps, and
    the disparity image equals the oracle's on exactly those images."""
    svmod = util.pkg("stereo_vision")
    rgb_l, rgb_r = util.load_png("kitti0_crop_color_left.png"), util.load_png("kitti0_crop_color_right.png")
    H, W = rgb_l.shape[:2]
    s = svmod.stereo_vision(objectTracking=False, width=W, height=H)
    s.sv.sv_legacy_set_rectify(1)
    try:
        s.generatePointCloud(rgb_l[..., ::-1], rgb_r[..., ::-1])
        dmap = s.last_disparity_u8()
        s.sv.sv_legacy_rectify_maps.restype = ctypes.POINTER(ctypes.c_float)
        maps = np.ctypeslib.as_array(s.sv.sv_legacy_rectify_maps(), shape=(4, H, W)).copy()
        gl, gr = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
        s.sv.sv_legacy_last_gray.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert s.sv.sv_legacy_last_gray(gl.ctypes.data, gr.ctypes.data) == 0
    finally:
        s.close()
        s.sv.sv_legacy_set_rectify(0)
    want_l, want_r = _remap_linear_u8(_gray_cv4(rgb_l), maps[0], maps[1]), _remap_linear_u8(_gray_cv4(rgb_r), maps[2], maps[3])
    assert (want_l != _gray_cv4(rgb_l)).mean() > 0.2  # the maps of this calibration at this size are not the identity
    assert np.array_equal(gl, want_l) and np.array_equal(gr, want_r)
    d1, _, _ = oracle.process(ElasParams.driver(255), gl, gr)
    assert np.array_equal(dmap, np.clip(np.rint(d1 * np.float32(4.0)), 0, 255).astype(np.uint8))


def _resize_linear_8uc4(src, dw, dh):
    """cv::resize(INTER_LINEAR) for 8-bit images as legacy_kernels.hip restates it (OpenCV 4.x generic path; exact 2x decimation =
    the INTER_AREA shortcut): float32 coordinate arithmetic, 11-bit coefficients, the two-step integer blend."""
    sh, sw = src.shape[:2]
    S = src.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        return ((S[0::2, 0::2] + S[0::2, 1::2] + S[1::2, 0::2] + S[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    f32 = np.float32

    def coeffs(n_dst, n_src, clamp):
        scale = 1.0 / (n_dst / n_src)
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(f32)).astype(f32)
        if clamp:
            lo, hi = i < 0, i >= n_src - 1
            f = np.where(lo | hi, f32(0), f)
            i = np.where(lo, 0, np.where(hi, n_src - 1, i))
        c0 = np.rint((f32(1) - f) * f32(2048)).astype(np.int64)
        c1 = np.rint(f * f32(2048)).astype(np.int64)
        return i, c0, c1

    sx, a0, a1 = coeffs(dw, sw, True)
    sy, b0, b1 = coeffs(dh, sh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    h0 = S[y0][:, sx] * a0[None, :, None] + S[y0][:, sx1] * a1[None, :, None]
    h1 = S[y1][:, sx] * a0[None, :, None] + S[y1][:, sx1] * a1[None, :, None]
    return ((((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)


def _bgra(rgb):
    return np.ascontiguousarray(np.concatenate([rgb[..., ::-1], np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2))


@pytest.mark.gpu
def test_reference_binding_with_fourteen_arguments():
    """The reference's own ctypes binding, entry for entry (stereo_vision/sv.py:167,180,189): 14 argtypes, the BGRA buffers passed as
    bytes, 14 arguments.  Arguments 15/16 of the C signature are then undefined - the library must not read them: the map is
    the full-resolution golden one (half-resolution mode can only come from sv_legacy_set_subsampling)."""
    from numpy.ctypeslib import ndpointer
    eng = util.pkg("engine")
    L, R = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
    H, W = L.shape
    lib = _lib(eng)
    lib.generatePointCloud.restype = ndpointer(dtype=ctypes.c_double, shape=(W * H, 3))
    lib.generatePointCloud.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool,
                                       ctypes.c_bool, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.clean.restype = None
    yml = os.path.join(os.path.dirname(eng.LIB_PATH), "stereo_vision", "data", "kitti_2011_09_26.yml").encode("utf-8")
    left = _bgra(np.repeat(L[:, :, None], 3, 2)).tobytes()
    right = _bgra(np.repeat(R[:, :, None], 3, 2)).tobytes()
    try:
        for _ in range(3):  # (whatever the unread slots hold, call after call)
            pts = lib.generatePointCloud(left, right, yml, W, H, True, False, False, False, 1, 1, b"src/yolo/yolov4-tiny.cfg", b"src/yolo/yolov4-tiny.weights", b"src/yolo/classes.txt")
            assert pts.shape == (W * H, 3)
            w, h = ctypes.c_int(), ctypes.c_int()
            lib.sv_legacy_last_dmap.restype = ctypes.POINTER(ctypes.c_ubyte)
            dmap = np.ctypeslib.as_array(lib.sv_legacy_last_dmap(ctypes.byref(w), ctypes.byref(h)), shape=(H, W)).copy()
            final = util.golden_npz("kitti0_d256")["final1"].reshape(H, W)
            assert np.array_equal(dmap, np.clip(np.rint(final * np.float32(4.0)), 0, 255).astype(np.uint8))
    finally:
        lib.clean()


@pytest.mark.gpu
@pytest.mark.parametrize("factor", ["2x", "1.25x", "0.8x"])
def test_frames_of_another_size_are_resized(oracle, factor):
    """stereo_vision.cpp:587-591: every call's width x height buffers are resized to the size the first call froze.  The gray images
    the matcher received equal gray_cv4(resize restatement) of the frame, and the map equals the oracle's on those images."""
    eng = util.pkg("engine")
    rgb_l, rgb_r = util.load_png("kitti0_crop_color_left.png"), util.load_png("kitti0_crop_color_right.png")
    H, W = rgb_l.shape[:2]  # the frozen size: 320 x 128
    sw, sh = {"2x": (2 * W, 2 * H), "1.25x": (W * 5 // 4, H * 5 // 4), "0.8x": (W * 4 // 5, 102)}[factor]
    from PIL import Image
    big_l = np.asarray(Image.fromarray(rgb_l).resize((sw, sh), Image.BICUBIC))  # any frame of the other size will do
    big_r = np.asarray(Image.fromarray(rgb_r).resize((sw, sh), Image.BICUBIC))
    lib = _lib(eng)
    lib.generatePointCloud.restype = ctypes.c_void_p
    lib.generatePointCloud.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_bool, ctypes.c_bool, ctypes.c_bool,
                                       ctypes.c_bool, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    lib.clean.restype = None
    lib.getColor.restype = ctypes.POINTER(ctypes.c_ubyte)
    yml = os.path.join(os.path.dirname(eng.LIB_PATH), "stereo_vision", "data", "kitti_2011_09_26.yml").encode("utf-8")
    a_l, a_r, b_l, b_r = _bgra(rgb_l), _bgra(rgb_r), _bgra(big_l), _bgra(big_r)
    try:
        assert lib.generatePointCloud(a_l.ctypes.data, a_r.ctypes.data, yml, W, H, True, False, False, False, 1, 1, b"", b"", b"")  # freezes W x H
        assert lib.generatePointCloud(b_l.ctypes.data, b_r.ctypes.data, yml, sw, sh, True, False, False, False, 1, 1, b"", b"", b"")
        gl, gr = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
        lib.sv_legacy_last_gray.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert lib.sv_legacy_last_gray(gl.ctypes.data, gr.ctypes.data) == 0
        lib.sv_legacy_last_dmap.restype = ctypes.POINTER(ctypes.c_ubyte)
        dmap = np.ctypeslib.as_array(lib.sv_legacy_last_dmap(None, None), shape=(H, W)).copy()
        colors = np.ctypeslib.as_array(lib.getColor(), shape=(H, W, 4)).copy()
    finally:
        lib.clean()
    want_l, want_r = _resize_linear_8uc4(b_l, W, H), _resize_linear_8uc4(b_r, W, H)
    assert np.array_equal(colors, want_l)  # left_img_OLD, what the viewer colours the cloud with
    assert np.array_equal(gl, _gray_cv4(want_l[..., 2::-1])) and np.array_equal(gr, _gray_cv4(want_r[..., 2::-1]))
    d1, _, _ = oracle.process(ElasParams.driver(255), gl, gr)
    assert np.array_equal(dmap, np.clip(np.rint(d1 * np.float32(4.0)), 0, 255).astype(np.uint8))


@pytest.mark.gpu
def test_disparity_to_u8_device():
    """sv_disparity_to_u8_device == leftdpf.convertTo(dmap, CV_8UC1, 4.0) (stereo_vision.cpp:316): round half to even, saturate;
    every alignment of the four-pixels-per-thread kernel, ties, negatives (-10 / -1 invalid marks), values beyond 63.75."""
    import torch
    eng = util.pkg("engine")
    rng = np.random.default_rng(5)
    base = np.concatenate([rng.uniform(-12, 80, 5003).astype(np.float32), np.arange(-4, 300, dtype=np.float32) / 8, np.array([-10, -1, 0, 63.75, 63.875, 64, 1e9, -1e9], np.float32)])
    for off in range(4):
        for n in (1, 2, 3, 4, 5, 1023, 1024, 1025, base.size - off):
            a = np.ascontiguousarray(base[off:off + n])
            t = torch.from_numpy(base.copy()).cuda()[off:off + n]
            out = eng.disparity_to_u8(t.contiguous() if off == 0 else t)  # a view at an odd offset stays contiguous: unaligned start
            torch.cuda.synchronize()
            want = np.clip(np.rint(a * np.float32(4.0)), 0, 255).astype(np.uint8)
            assert np.array_equal(out.cpu().numpy(), want), (off, n)
