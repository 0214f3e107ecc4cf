"""CPU: the C-ABI library loads and exports every symbol include/stereo_vision_hip.h declares; the product's
host-side stages (which run on the CPU by design) match the oracle; argument validation fails loudly."""
import ctypes
import os
import re

import numpy as np
import pytest

import util
from pyoracle import ElasParams

HEADER = os.path.join(util.ROOT, "include", "stereo_vision_hip.h")


@pytest.fixture(scope="module")
def eng():
    util.pkg("build").build()
    return util.pkg("engine")


def test_header_symbols_exported(eng):
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(sv_[a-z_]+|generatePointCloud|clean|getColor)\s*\(", src))
    assert {"sv_create", "sv_process_batch_device", "generatePointCloud", "clean", "getColor"} <= names
    eng.share_hip_runtime_with_torch()
    L = ctypes.CDLL(eng.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_params_presets_match_reference_values(eng):
    for setting in ("robotics", "middlebury", "driver"):
        a = eng.SvParams.preset(setting)
        b = ElasParams.driver(255) if setting == "driver" else ElasParams.preset(setting)
        for f, _ in eng.SvParams._fields_:
            assert getattr(a, f) == getattr(b, f), (setting, f)


def test_create_rejects_unsupported(eng):
    L = eng.lib()
    cfg = eng.SvConfig(1242, 375, 0, 1, 1, 0)
    h = ctypes.c_void_p()
    p = eng.SvParams.driver(5)
    assert L.sv_create(ctypes.byref(p), ctypes.byref(cfg), ctypes.byref(h)) == -1 and not h.value
    assert b"disp_max" in L.sv_last_error(None)
    assert L.sv_create(None, None, None) == -1
    # policy words are range-checked before anything touches a device: one past the last policy of some of them, a reserved word
    p = eng.SvParams.driver(127)
    for field, bad in (("latency_split", 4), ("gpu_triangulation", 5), ("host_copies", 3), ("event_sync", 4), ("gpu_triangulation_pct", 101), ("affinity", 3), ("latency_split", -1)):
        cfg = eng.SvConfig(1242, 375, 0, 1, 1, 0)
        setattr(cfg, field, bad)
        assert L.sv_create(ctypes.byref(p), ctypes.byref(cfg), ctypes.byref(h)) == -1 and not h.value, field
        assert field.encode() in L.sv_last_error(None), (field, L.sv_last_error(None))
    cfg = eng.SvConfig(1242, 375, 0, 1, 1, 0)
    cfg.reserved[3] = 1
    assert L.sv_create(ctypes.byref(p), ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"reserved" in L.sv_last_error(None)


def test_no_gpu_fails_loudly(eng):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(eng.StereoError):
        eng.StereoEngine(320, 120, eng.SvParams.driver(63))


@pytest.mark.parametrize("name", ["kitti0_d128", "kitti0_crop_d64", "cones_crop_robotics", "cones_crop_middlebury", "synth7_d64", "synth8_d32"])
def test_host_stage_matches_oracle(eng, oracle, name):
    e = util.digests()[name]
    L, R = util.case_images(e)
    oracle.run_stages(util.case_params(e, ElasParams), L, R)
    s = eng.host_support_filter(util.case_params(e, eng.SvParams), oracle.stage("dcan_raw"), L.shape[1], L.shape[0])
    assert np.array_equal(s.ravel(), oracle.stage("support"))
    s3 = eng.host_support_filter(util.case_params(e, eng.SvParams), oracle.stage("dcan_raw"), L.shape[1], L.shape[0], threads=3)
    assert np.array_equal(s3, s)
    tl = eng.host_delaunay(np.stack([s[:, 0], s[:, 1]], 1))
    tr = eng.host_delaunay(np.stack([s[:, 0] - s[:, 2], s[:, 1]], 1))
    assert np.array_equal(tl.ravel(), oracle.stage("tri1")) and np.array_equal(tr.ravel(), oracle.stage("tri2"))


def test_host_delaunay_random_sets(eng, oracle):
    rng = np.random.default_rng(5)
    for it in range(300):
        n = int(rng.integers(3, 250))
        if it % 3 == 0:
            pts = rng.integers(0, 60, (n, 2)) * 5
        elif it % 3 == 1:
            pts = np.stack([rng.integers(-50, 300, n), rng.integers(0, 75, n) * 5], 1)
        else:
            pts = np.stack([rng.integers(0, 8, n) * 5, rng.integers(0, 8, n) * 5], 1)
        if len(np.unique(pts, axis=0)) < 3:
            continue
        a, b = oracle.delaunay(pts.astype(np.float32)), eng.host_delaunay(pts)
        assert a.shape == b.shape and np.array_equal(a, b), it


def test_host_filter_random_lattices(eng, oracle):
    """The vectorised lattice filters (mask-driven scans, 16 rows per step) against the restatement's plain loops on random
    lattices: sizes that are not multiples of 16, sparse and dense, every window size, with and without corner points."""
    L = oracle.lib
    L.orc_support_filter.argtypes = [ctypes.POINTER(ElasParams), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.orc_support_filter.restype = ctypes.c_int
    rng = np.random.default_rng(9)
    for it in range(80):
        W, H = int(rng.integers(40, 700)), int(rng.integers(40, 400))
        po = ElasParams.preset("robotics" if it % 2 else "middlebury")
        po.candidate_stepsize = int(rng.choice([3, 5, 6, 8]))
        po.incon_window_size = int(rng.choice([0, 1, 3, 5, 7]))
        po.incon_threshold = int(rng.integers(1, 8))
        po.incon_min_support = int(rng.integers(1, 12))
        po.add_corners = int(rng.integers(0, 2))
        pe = eng.SvParams.preset("robotics")
        for f, _ in eng.SvParams._fields_:
            setattr(pe, f, getattr(po, f))
        step = po.candidate_stepsize
        Wc, Hc = (W + step - 1) // step, (H + step - 1) // step
        noise = rng.integers(0, 60, (Hc, Wc))
        ramp = np.add.outer(np.arange(Hc), np.arange(Wc)) // int(rng.integers(2, 9)) + int(rng.integers(0, 30))
        d = np.where(rng.random((Hc, Wc)) < 0.5, noise, ramp)
        d = np.where(rng.random((Hc, Wc)) < rng.uniform(0.05, 0.9), d, -1).astype(np.int16)
        d[0, :] = 0
        d[:, 0] = 0
        want = np.zeros((Wc * Hc + 6, 3), np.int32)
        a = d.copy()
        n = L.orc_support_filter(ctypes.byref(po), a.ctypes.data, W, H, want.ctypes.data, want.shape[0])
        got = eng.host_support_filter(pe, d.copy(), W, H)
        assert n == got.shape[0] and np.array_equal(got, want[:n]), (it, W, H, step)
        # the same lattice shared between the threads of a team (what single-pair calls do): same list, same filtered lattice
        for threads in (2, 5):
            shared, lat = eng.host_support_filter(pe, d.copy(), W, H, threads=threads, lattice=True)
            assert np.array_equal(shared, want[:n]) and np.array_equal(lat, a), (it, W, H, step, threads)


def test_host_delaunay_split_halves(eng):
    """Latency mode builds the two halves of the top-level cut on two threads into pre-computed slot ranges: same triangles,
    same order - also when the helper comes too late and the caller builds both halves itself."""
    rng = np.random.default_rng(17)
    for it in range(60):
        n = int(rng.integers(64, 3000))
        if it % 3 == 0:
            pts = rng.integers(0, 250, (n, 2)) * 5
        elif it % 3 == 1:
            pts = np.stack([rng.integers(-50, 1300, n), rng.integers(0, 75, n) * 5], 1)
        else:
            pts = rng.integers(0, 4000, (n, 2))
        a = eng.host_delaunay(pts)
        b = eng.host_delaunay(pts, split=True, helper_delay_us=0 if it % 2 else 20000)
        assert a.shape == b.shape and np.array_equal(a, b), it
        c = eng.host_delaunay(pts, split=True, depth=2 + it % 2, helper_delay_us=0 if it % 4 < 2 else 3000)  # quarters / eighths
        assert a.shape == c.shape and np.array_equal(a, c), it
