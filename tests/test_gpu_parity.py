"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle on the same inputs and against the
golden vectors the compiled reference produced.  Bit-exact at every stage (integer WTA index AND the float maps
after gap interpolation / adaptive mean / median — tolerance 0)."""
import ctypes

import numpy as np
import pytest

import util
from pyoracle import ElasParams

pytestmark = pytest.mark.gpu

DIG = util.digests()
STAGES = ["desc1", "desc2", "dcan_raw", "support", "tri1", "tri2", "planes1", "planes2", "grid1", "grid2",
          "wta1", "wta2", "lr1", "lr2", "speckle1", "speckle2", "gap1", "gap2", "amean1", "amean2", "final1", "final2"]


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback"
    return util.pkg("engine")


def _run_debug(eng, entry, gpu_filter=None, debug=None):
    L, R = util.case_images(entry)
    p = util.case_params(entry, eng.SvParams)
    e = eng.StereoEngine(L.shape[1], L.shape[0], p, keep_debug=True, gpu_filter=gpu_filter)
    try:
        for key, value in (debug or {}).items():
            e.debug_set(key, value)
        d1, d2, status = e.process_host(L, R)
        stages = {}
        for k in STAGES:
            try:
                stages[k] = e.debug(k)
            except KeyError:
                pass
    finally:
        e.close()
    return L, R, d1[0], d2[0], int(status[0]), stages


@pytest.mark.parametrize("gpu_filter", [False, True])
@pytest.mark.parametrize("name", sorted(DIG))
def test_every_stage_matches_oracle_and_golden(eng, oracle, name, gpu_filter):
    """gpu_filter: lattice filters on the GPU (k_support_filter, the throughput configuration) or on the host pool."""
    entry = DIG[name]
    L, R, d1, d2, nsup, st = _run_debug(eng, entry, gpu_filter=True if gpu_filter else None)
    assert nsup == entry["n_support"]
    oracle.run_stages(util.case_params(entry, ElasParams), L, R)
    bad = []
    for k in STAGES:
        o = oracle.stage(k)
        g = st.get(k)
        if g is None or g.size != o.size or not np.array_equal(g.view(np.uint8), o.view(np.uint8)):
            nd = int((g != o).sum()) if g is not None and g.size == o.size else -1
            bad.append((k, nd))
        elif util.sha(g) != entry["stages"][k]:
            bad.append((k, "golden digest"))
    assert not bad, "HIP stages differ: %s" % bad
    # the caller-visible maps are the final left map and the L/R-checked (or fully post-processed) right map
    assert util.sha(d1) == entry["stages"]["final1"]
    assert util.sha(d2) == entry["stages"]["final2"]


def test_batch_of_distinct_pairs_many_workers(eng, oracle):
    """B distinct pairs through several workers/streams and chunks: every pair equals its own oracle result."""
    import torch
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 13
    batch = synth.make_batch(200, B, H, W, D)
    p = eng.SvParams.driver(D - 1)
    e = eng.StereoEngine(W, H, p, n_workers=3, chunk=2, n_streams=2, n_slots=3)
    try:
        left = torch.from_numpy(batch[:, 0].copy()).cuda()
        right = torch.from_numpy(batch[:, 1].copy()).cuda()
        status = np.zeros(B, np.int32)
        d1, d2 = e.process_device(left, right, status=status)
        d1b, d2b = e.process_device(left, right)  # second call on the same handle: buffers are reused
        torch.cuda.synchronize()
        assert torch.equal(d1, d1b) and torch.equal(d2, d2b)
        d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    finally:
        e.close()
    po = ElasParams.driver(D - 1)
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)), "pair %d D1" % i
        assert np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), "pair %d D2" % i
        assert status[i] >= 3


def test_textureless_and_too_few_support_points(eng, oracle):
    """All-zero images (the reference's own smoke input, tests/test_demo.py:8-10).  Driver preset: only the six corner
    points; ROBOTICS: <3 support points -> maps left untouched (elas.cpp:63-69), status reports it."""
    Z = np.zeros((80, 200), np.uint8)
    e = eng.StereoEngine(200, 80, eng.SvParams.driver(63))
    try:
        d1, d2, status = e.process_host(Z, Z)
    finally:
        e.close()
    o1, o2, _ = oracle.process(ElasParams.driver(63), Z, Z)
    assert np.array_equal(d1[0], o1) and np.array_equal(d2[0], o2) and status[0] == 6
    p = eng.SvParams.preset("robotics")
    p.disp_max = 63
    e = eng.StereoEngine(200, 80, p)
    try:
        d1, d2, status = e.process_host(Z, Z)
    finally:
        e.close()
    assert status[0] < 3 and not d1.any() and not d2.any()


@pytest.mark.parametrize("gpu_filter", [False, True])
@pytest.mark.parametrize("W,H,D", [(32, 32, 16), (47, 33, 16), (65, 40, 24), (130, 36, 32), (257, 67, 48), (513, 35, 64), (96, 131, 32), (4100, 83, 64), (121, 1101, 32)])
def test_small_and_odd_image_sizes(eng, oracle, W, H, D, gpu_filter, monkeypatch):
    """Image sizes nothing is tuned for: the smallest the library takes (32 x 32: a 7 x 7 lattice, two grid cells a side), widths that are
    no multiple of 4 / 64 / the tile widths, a map narrower than one tile and one a single column wider than a tile, more rows than
    columns, a row of more than 4 096 pixels, a column of more than 1 024 - five distinct seeded pairs each through a chunk-4 pipeline (a ragged last chunk), both presets, lattice filters on
    the GPU and on the host: every map equals the oracle's.  (The compiled reference agrees with the oracle on the 257 x 67 and
    96 x 131 cases and segfaults on the five with at most 40 rows - two grid-cell rows leave its flat 3 x 3 dilation, elas.cpp:613-628,
    without a valid range; there the memory-safe restatement is what defines the result.)"""
    synth = util.pkg("synth")
    batch = synth.make_batch(9000 + W, 5, H, W, D)
    for preset in ("driver", "robotics"):
        p = eng.SvParams.driver(D - 1) if preset == "driver" else eng.SvParams.preset("robotics")
        po = ElasParams.driver(D - 1) if preset == "driver" else ElasParams.preset("robotics")
        p.disp_max = po.disp_max = D - 1
        e = eng.StereoEngine(W, H, p, chunk=4, n_slots=2, n_workers=3, gpu_filter=bool(gpu_filter))
        try:
            d1, d2, status = e.process_host(np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1]))
        finally:
            e.close()
        for i in range(batch.shape[0]):
            o1 = np.zeros((H, W), np.float32)
            o2 = np.zeros((H, W), np.float32)
            n = oracle.run_stages(po, batch[i, 0], batch[i, 1])
            if n >= 3:  # (fewer: the maps stay as the caller handed them over, elas.cpp:63-69 - zeros here)
                o1, o2 = oracle.stage("final1").reshape(H, W), oracle.stage("final2").reshape(H, W)
            assert status[i] == n, (preset, i, status[i], n)
            assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)), (preset, i, "D1")
            assert np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), (preset, i, "D2")


def test_elas_process_seam(eng, oracle):
    """sv_elas_process has Elas::process's argument meaning (elas.h:153-162)."""
    entry = DIG["kitti0_crop_d64"]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams))
    try:
        D1, D2 = e.elas_process(L, R)
    finally:
        e.close()
    assert util.sha(D1) == entry["stages"]["final1"] and util.sha(D2) == entry["stages"]["final2"]


def test_determinism_full_size(eng):
    """KITTI-size batch twice (different worker/chunk geometry): identical bytes; and size-independent properties of the
    maps: values are -10 or in [0, disp_max], right map only holds integers or -10 (postprocess_only_left)."""
    import torch
    synth = util.pkg("synth")
    B = 6
    batch = synth.make_batch(1000, B)
    left = torch.from_numpy(batch[:, 0].copy()).cuda()
    right = torch.from_numpy(batch[:, 1].copy()).cuda()
    outs = []
    for nw, ch, ns, nsl in ((1, 6, 1, 1), (4, 1, 2, 4)):
        e = eng.StereoEngine(1242, 375, eng.SvParams.driver(127), n_workers=nw, chunk=ch, n_streams=ns, n_slots=nsl)
        try:
            d1, d2 = e.process_device(left, right)
            torch.cuda.synchronize()
            outs.append((d1.cpu().numpy(), d2.cpu().numpy()))
        finally:
            e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    d1, d2 = outs[0]
    assert ((d1 == -10) | ((d1 >= 0) & (d1 <= 127))).all()
    assert ((d2 == -10) | ((d2 >= 0) & (d2 <= 127) & (d2 == np.round(d2)))).all()
    assert (d1 >= 0).mean() > 0.9


def test_raster_fallback_path(eng, oracle, monkeypatch):
    """Tile lists that overflow switch a map to the global-atomic rasteriser: force that with a tiny cap and check the maps."""
    entry = DIG["kitti0_crop_d64"]
    L, R, d1, d2, nsup, st = _run_debug(eng, entry, debug={"rt_cap": 3})
    assert util.sha(st["wta1"]) == entry["stages"]["wta1"] and util.sha(st["wta2"]) == entry["stages"]["wta2"]
    assert util.sha(d1) == entry["stages"]["final1"] and util.sha(d2) == entry["stages"]["final2"]


def test_adaptive_mean_division_is_exact(eng):
    """k_amean divides by v_rcp_f32 + one FMA correction instead of the 10-instruction IEEE sequence.  The divisor (a sum of eight
    weights 0 / 2 / 4) is an even integer in [2, 32]; for those sixteen values the shortcut must give the correctly rounded
    quotient of EVERY float: compared here on the device for all 2^23 mantissas, both signs, 31 exponents (2^-15 .. 2^16)."""
    import ctypes
    L = eng.lib()
    L.sv_debug_check_amean_div.restype = ctypes.c_longlong
    L.sv_debug_check_amean_div.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_longlong)]
    a, d, control = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_longlong(0)
    bad = L.sv_debug_check_amean_div(ctypes.byref(a), ctypes.byref(d), ctypes.byref(control))
    assert bad == 0, (bad, hex(a.value), hex(d.value))
    assert control.value > 0  # a * rcp(d) alone is NOT always the rounded quotient: the comparison does discriminate


def test_gpu_delaunay_matches_host(eng):
    """csrc/delaunay_gpu.hip (the divide-and-conquer phase as one workgroup per vertex set, mesh in LDS) against the host
    triangulation: random lattice / scattered / heavily co-circular sets up to the kernel's limit, and the support points of the
    kitti_mini pair in both images; several sets in one launch."""
    rng = np.random.default_rng(41)
    for it in range(60):
        n = int(rng.integers(3, 3900 if it % 10 == 0 else 500))
        k = it % 4
        if k == 0:
            pts = rng.integers(0, 250, (n, 2)) * 5
        elif k == 1:
            pts = np.stack([rng.integers(-50, 1300, n), rng.integers(0, 75, n) * 5], 1)
        elif k == 2:
            pts = np.stack([rng.integers(0, 12, n) * 5, rng.integers(0, 12, n) * 5], 1)
        else:
            pts = rng.integers(0, 3000, (n, 2))
        if len(np.unique(pts, axis=0)) < 3:
            continue
        a, (b, _) = eng.host_delaunay(pts), eng.gpu_delaunay(pts)
        assert a.shape == b.shape and np.array_equal(a, b), (it, n)
    g = util.golden_npz("kitti0_d128")
    s = g["support"].reshape(-1, 3)
    for side, want in ((0, g["tri1"]), (1, g["tri2"])):
        pts = np.stack([s[:, 0] - (s[:, 2] if side else 0), s[:, 1]], 1)
        got, _ = eng.gpu_delaunay(pts, reps=16)
        assert np.array_equal(got.ravel(), want.ravel())


def test_gpu_delaunay_large_sets_match_host(eng, monkeypatch):
    """Sets beyond the LDS limit (k_dgl_subtrees / k_dgl_top: subtrees of the recursion in LDS, the upper merges in a global-memory
    mesh): 4K-sized lattices of 5 000 - 60 000 points, several sets per launch; and, with the subtree limit lowered, cuts down to
    depth 6 on small sets (duplicates, co-circular lattices, scattered points)."""
    rng = np.random.default_rng(43)
    for it, n in enumerate([4001, 5000, 9000, 17000, 31000, 60000]):
        if it % 2 == 0:
            pts = np.stack([rng.integers(0, 768, n) * 5, rng.integers(0, 432, n) * 5], 1)
        else:
            pts = np.stack([rng.integers(-190, 3840, n), rng.integers(0, 432, n) * 5], 1)
        a, (b, ms) = eng.host_delaunay(pts), eng.gpu_delaunay(pts, reps=1 if n > 20000 else 3)
        assert a.shape == b.shape and np.array_equal(a, b), n
    for sub in (6, 7, 50, 333):
        monkeypatch.setenv("SV_DG_SUBMAX", str(sub))
        for it in range(12):
            n = int(rng.integers(sub + 1, sub * 64 + 1))
            k = it % 3
            if k == 0:
                pts = rng.integers(0, 60, (n, 2)) * 5
            elif k == 1:
                pts = np.stack([rng.integers(-50, 1300, n), rng.integers(0, 75, n) * 5], 1)
            else:
                pts = rng.integers(0, 3000, (n, 2))
            m = len(np.unique(pts, axis=0))
            if m < 3 or (m + 63) // 64 > sub:
                continue
            a, (b, _) = eng.host_delaunay(pts), eng.gpu_delaunay(pts, reps=2)
            assert a.shape == b.shape and np.array_equal(a, b), (sub, it, n)


@pytest.mark.parametrize("name", ["kitti0_d128", "kitti20_d128", "kitti0_d128_sub", "cones_crop_middlebury", "synth5000_4kstrip_d192"])
def test_pipeline_with_gpu_triangulation(eng, oracle, monkeypatch, name):
    """SV_GPU_DELAUNAY=1 (what a handle with few host threads chooses by itself): the host pool only orders the vertices, the
    triangle lists are built on the device straight into the chunk's blob.  Same maps, bit for bit; the 4K strip's 7 500-point
    sets exceed the kernel's LDS and take the cut path (subtrees in LDS, upper merges in a global-memory mesh)."""
    entry = DIG[name]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=3, triangulation="gpu")
    try:
        assert e.query()["gpu_triangulation"] == 1
        d1, d2, st = e.process_host(np.stack([L] * 5), np.stack([R] * 5))
    finally:
        e.close()
    assert (st == entry["n_support"]).all()
    for i in range(5):
        assert util.sha(d1[i]) == entry["stages"]["final1"]
        assert util.sha(d2[i]) == entry["stages"]["final2"]


@pytest.mark.parametrize("sub_max", ["700", "40"])
def test_pipeline_with_cut_triangulation(eng, oracle, monkeypatch, sub_max):
    """The cut path of the GPU triangulation inside the pipeline on ordinary images: with the LDS limit lowered to 700 (40)
    vertices the 2 100-point sets of a KITTI pair are built as 4 (64) subtrees plus the upper merges in global memory."""
    entry = DIG["kitti0_d128"]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=3, triangulation="gpu", dg_sub_max=int(sub_max))
    try:
        d1, d2, st = e.process_host(np.stack([L] * 6), np.stack([R] * 6))
        assert e.gpu_triangulation_fallbacks() == 0
    finally:
        e.close()
    assert (st == entry["n_support"]).all()
    for i in range(6):
        assert util.sha(d1[i]) == entry["stages"]["final1"] and util.sha(d2[i]) == entry["stages"]["final2"]


def test_gpu_triangulation_falls_back_per_set(eng, oracle, monkeypatch):
    """Vertex sets beyond what the GPU kernels take (here capped at 5 000 points: the 4K strip has ~7 500) are triangulated by the
    pool inside a chunk whose other work stays on the GPU; the handle counts them (SV_Q_GPU_TRIANGULATION_FALLBACKS)."""
    entry = DIG["synth5000_4kstrip_d192"]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=3, triangulation="gpu", dg_max_points=5000)
    try:
        d1, d2, st = e.process_host(np.stack([L] * 5), np.stack([R] * 5))
        assert e.gpu_triangulation_fallbacks() == 10  # both sides of the five pairs
    finally:
        e.close()
    assert (st == entry["n_support"]).all() and entry["n_support"] > 5000
    for i in range(5):
        assert util.sha(d1[i]) == entry["stages"]["final1"] and util.sha(d2[i]) == entry["stages"]["final2"]


@pytest.mark.parametrize("resident", [True, False])
def test_sets_beyond_the_cut_table_go_to_the_host(eng, resident):
    """Round 4's fault, pinned: a set of more than dg_sub_max << 6 vertices needs a deeper cut than a set's node-result table holds
    (delaunay_gpu.hip: DG_CUT_MAX).  With dg_sub_max = 100 the 4K strip's 7 528-vertex sets (limit 6 400) must be handed to the host -
    by the handle's clamp (engine.cpp: dg_limit), the launchers' check and the kernels' own skip - and every map must still be the
    reference's.  Asked for with and without resident chunks."""
    entry = DIG["synth5000_4kstrip_d192"]
    assert entry["n_support"] > 100 << 6
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=3, triangulation="gpu", dg_sub_max=100,
                         resident=None if resident else False)
    try:
        # (a handle whose GPU limit is below the lists its bulk copy can hold does not go resident at all: sv_create's own rule)
        assert e.query()["gpu_triangulation"] == 1 and e.query()["resident"] == 0
        d1, d2, st = e.process_host(np.stack([L] * 5), np.stack([R] * 5))
        assert e.gpu_triangulation_fallbacks() == 10  # both sides of the five pairs went to the pool, none into the node-result table
    finally:
        e.close()
    assert (st == entry["n_support"]).all()
    for i in range(5):
        assert util.sha(d1[i]) == entry["stages"]["final1"] and util.sha(d2[i]) == entry["stages"]["final2"]


def test_cut_launcher_refuses_sets_it_cannot_cut(eng, monkeypatch):
    """The launchers themselves return an error for a set that needs more than 2^6 subtrees instead of launching a smaller grid."""
    rng = np.random.default_rng(5)
    monkeypatch.setenv("SV_DG_SUBMAX", "100")
    lat = rng.permutation(120 * 75)[:7000]
    pts = np.stack([(lat % 120) * 5, (lat // 120) * 5], 1)  # 7 000 distinct lattice points > 100 << 6
    with pytest.raises(eng.StereoError):
        eng.gpu_delaunay(pts)
    got, _ = eng.gpu_delaunay(pts[:6400])  # exactly at the limit: depth 6
    assert np.array_equal(got, eng.host_delaunay(pts[:6400]))


def test_pipeline_with_mixed_triangulation(eng, oracle, monkeypatch):
    """SV_GPU_DELAUNAY_PCT=40: inside one chunk some pairs are triangulated by the pool, the others by the GPU kernel (per-pair
    flag in the blob's meta words); every pair still equals its own oracle result."""
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 11
    batch = synth.make_batch(260, B, H, W, D)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=4, n_slots=2, n_streams=2, n_workers=3, triangulation=40)
    try:
        assert e.query()["gpu_triangulation"] == 0
        d1, d2, st = e.process_host(batch[:, 0], batch[:, 1])
    finally:
        e.close()
    po = ElasParams.driver(D - 1)
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        assert st[i] >= 3
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i


def test_pipeline_with_balanced_triangulation(eng, oracle, monkeypatch):
    """Host mode with a pool that cannot keep up (2 threads): the dispatcher hands a growing share of each chunk to the GPU
    triangulation kernel (engine.cpp dispatcher_main); which pairs it takes depends on timing, the maps do not."""
    synth = util.pkg("synth")
    H, W, D, B = 375, 1242, 128, 4  # full-size pairs: two threads triangulate ~5 000 of them per second, the GPU asks for far more
    batch = synth.make_batch(411, B, H, W, D)
    l, r = np.concatenate([batch[:, 0]] * 32), np.concatenate([batch[:, 1]] * 32)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=8, n_slots=6, n_streams=2, n_workers=2, triangulation="balanced")
    try:
        assert e.query()["gpu_triangulation"] == 0
        d1, d2, st = e.process_host(l, r)
        share = e.gpu_triangulation_share()
    finally:
        e.close()
    po = ElasParams.driver(D - 1)
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        for rep in range(32):
            q = i + rep * B
            assert st[q] >= 3
            assert np.array_equal(d1[q].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[q].view(np.uint8), o2.view(np.uint8)), q
    if share == 0:
        pytest.skip("the pool never fell behind on this box: no pair went to the GPU kernel")


def test_random_parameter_sets(eng, oracle):
    """Elas::parameters far from the three presets (tools/fuzz_params.py): both maps, batch path and latency path, bit-exact."""
    import sys

    sys.path.insert(0, util.ROOT + "/tools")
    import fuzz_params as fz

    rng = np.random.default_rng(33)
    shapes = [(150, 260), (97, 203), (200, 320), (128, 401)]
    for i in range(16):
        vals = fz.random_params(rng)
        res = fz.run_case(eng, oracle, util.pkg("synth"), vals, 700 + i, shapes[i % 4])
        assert all(r[0] and r[1] for r in res), (res, vals)


@pytest.mark.parametrize("inline", [True, False])
def test_latency_mode_single_pairs(eng, monkeypatch, inline):
    """chunk = 1, one pair per call: the calling thread drives the pair itself (run_inline); SV_NO_INLINE sends it through the
    queued pipeline instead.  Both must give the golden maps, also for a pair without support points in between."""
    entry = DIG["kitti0_d128"]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=1, n_slots=2, n_streams=1, n_workers=4, inline=None if inline else False)  # 4 workers: split triangulations
    try:
        for rep in range(3):
            d1, d2, st = e.process_host(L, R)
            assert st[0] == entry["n_support"] and util.sha(d1[0]) == entry["stages"]["final1"]
            z1, z2, zs = e.process_host(np.zeros_like(L), np.zeros_like(R))
            assert zs[0] >= 3 or not z1.any()  # MIDDLEBURY adds corner points: either a flat map or an untouched (zero) one
    finally:
        e.close()


@pytest.mark.parametrize("cap", ["8", "300"])
def test_speckle_slow_path(eng, oracle, monkeypatch, cap):
    """Bands with more runs than the LDS run tables hold switch their map to the per-pixel union-find: force that with tiny
    tables (8: every map; 300: only some bands overflow) and check the speckle stage and the final maps."""
    for name in ("kitti0_crop_d64", "kitti0_d128"):
        entry = DIG[name]
        L, R, d1, d2, nsup, st = _run_debug(eng, entry, debug={"ccl_cap": int(cap)})
        assert util.sha(st["speckle1"]) == entry["stages"]["speckle1"]
        assert util.sha(d1) == entry["stages"]["final1"]


@pytest.mark.parametrize("gpu_filter", [False, True])
def test_full_4k_pair(eng, oracle, gpu_filter, monkeypatch):
    """BASELINE config 5 shape: one full 3840x2160 synthetic pair at D=192 against the oracle (bit-exact); with the lattice
    filters on the host pool and on the GPU (a 768 x 432 lattice: the multi-kernel filter has no size limit)."""
    synth = util.pkg("synth")
    L, R = synth.make_pair(5001, 2160, 3840, 192, scale=3)
    p = eng.SvParams.driver(191)
    e = eng.StereoEngine(3840, 2160, p, chunk=1, n_slots=1, n_streams=1, n_workers=2, gpu_filter=True if gpu_filter else None)
    try:
        assert e.query()["gpu_lattice_filter"] == int(gpu_filter)
        d1, d2, status = e.process_host(L, R)
    finally:
        e.close()
    o1, o2, _ = _oracle_4k(oracle, L, R)
    assert status[0] >= 3
    assert np.array_equal(d1[0].view(np.uint8), o1.view(np.uint8))
    assert np.array_equal(d2[0].view(np.uint8), o2.view(np.uint8))


def test_full_4k_batch_without_host_triangulation(eng, oracle, monkeypatch):
    """Config 5 with everything between the two kernel phases on the GPU: lattice filters, the vertex orders of the 21 000-point sets
    (k_dg_prepare_large_blob: bit maps and alternating cuts in the slot's global-memory scratch) and their triangulations (subtrees in
    LDS + upper merges in a global-memory mesh); the support lists never leave the device, the two host threads read 8 meta words per
    pair.  No set falls back to the host; maps bit-exact."""
    synth = util.pkg("synth")
    L, R = synth.make_pair(5001, 2160, 3840, 192, scale=3)
    e = eng.StereoEngine(3840, 2160, eng.SvParams.driver(191), chunk=4, n_slots=2, n_streams=2, n_workers=2, gpu_filter=True, triangulation="gpu")
    try:
        q = e.query()
        assert q["gpu_lattice_filter"] == 1 and q["gpu_triangulation"] == 1 and q["resident"] == 1  # (the vertex orders too: k_dg_prepare_large_blob)
        d1, d2, status = e.process_host(np.stack([L] * 5), np.stack([R] * 5))
        assert e.gpu_triangulation_share() == 1.0 and e.gpu_triangulation_fallbacks() == 0
    finally:
        e.close()
    o1, o2, _ = _oracle_4k(oracle, L, R)
    assert (status > 4000).all()  # beyond the LDS kernel
    for i in range(5):
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)), i
        assert np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i


_ORACLE_4K = {}


def _oracle_4k(oracle, L, R):
    if "r" not in _ORACLE_4K:  # the oracle needs ~1.5 s for a 4K pair: once for both parametrisations
        _ORACLE_4K["r"] = oracle.process(ElasParams.driver(191), L, R)
    return _ORACLE_4K["r"]


def test_robotics_preset_batch(eng, oracle):
    """ROBOTICS preset (texture gate, gap width 3, no median, no corner points) on a small batch."""
    import torch
    synth = util.pkg("synth")
    H, W, D, B = 96, 256, 48, 5
    batch = synth.make_batch(300, B, H, W, D)
    p = eng.SvParams.preset("robotics")
    p.disp_max = D - 1
    e = eng.StereoEngine(W, H, p, chunk=2, n_slots=2)
    try:
        d1, d2 = e.process_device(torch.from_numpy(batch[:, 0].copy()).cuda(), torch.from_numpy(batch[:, 1].copy()).cuda())
        torch.cuda.synchronize()
        d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    finally:
        e.close()
    po = ElasParams.preset("robotics")
    po.disp_max = D - 1
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i


@pytest.mark.parametrize("disp_max", [300, 511])
def test_wide_disparity_range(eng, oracle, disp_max):
    """disp_max beyond 255: more than eight candidate-mask words per cell in the dense kernel, and (511) support matching with
    more than 64 KB of LDS per workgroup (hipFuncSetAttribute path)."""
    import torch
    synth = util.pkg("synth")
    H, W, B = 90, 700, 3
    batch = synth.make_batch(4100, B, H, W, 64)
    p = eng.SvParams.driver(disp_max)
    e = eng.StereoEngine(W, H, p, chunk=2, n_slots=2)
    try:
        d1, d2 = e.process_device(torch.from_numpy(batch[:, 0].copy()).cuda(), torch.from_numpy(batch[:, 1].copy()).cuda())
        torch.cuda.synchronize()
        d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    finally:
        e.close()
    for i in range(B):
        o1, o2, _ = oracle.process(ElasParams.driver(disp_max), batch[i, 0], batch[i, 1])
        assert (o1 >= 0).mean() > 0.3  # a real match, not the "too few support points" early exit
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i


def test_streamed_submission_equals_synchronous(eng):
    """sv_submit_batch_device x3 + sv_wait gives the same bytes as three synchronous calls (distinct output buffers)."""
    import torch
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 10
    batches = [synth.make_batch(400 + 50 * i, B, H, W, D) for i in range(3)]
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=4, n_slots=3, n_workers=4)
    try:
        ins = [(torch.from_numpy(b[:, 0].copy()).cuda(), torch.from_numpy(b[:, 1].copy()).cuda()) for b in batches]
        ref = [e.process_device(l, r) for l, r in ins]
        outs = [(torch.zeros_like(ref[0][0]), torch.zeros_like(ref[0][1])) for _ in ins]
        torch.cuda.synchronize()
        for (l, r), (o1, o2) in zip(ins, outs):
            e.submit_device(l, r, o1, o2)
        e.wait()
        torch.cuda.synchronize()
        for (a1, a2), (b1, b2) in zip(ref, outs):
            assert torch.equal(a1, b1) and torch.equal(a2, b2)
    finally:
        e.close()


def test_wait_batches_hands_over_finished_batches_in_order(eng):
    """sv_wait_batches(n): the n oldest submitted batches are complete (their maps final) while later ones may still run; the
    consumer of bench.py's chunked gather relies on it.  Device and host-memory batches."""
    import torch
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 6
    batches = [synth.make_batch(700 + 20 * i, B, H, W, D) for i in range(4)]
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=4, n_slots=3, n_workers=4)
    try:
        ins = [(torch.from_numpy(b[:, 0].copy()).cuda(), torch.from_numpy(b[:, 1].copy()).cuda()) for b in batches]
        ref = [e.process_device(l, r) for l, r in ins]
        outs = [(torch.full_like(ref[0][0], -77.0), torch.full_like(ref[0][1], -77.0)) for _ in ins]
        torch.cuda.synchronize()
        for (l, r), (o1, o2) in zip(ins, outs):
            e.submit_device(l, r, o1, o2)
        for i in range(len(ins)):
            e.wait_batches(i + 1)
            torch.cuda.synchronize()
            assert torch.equal(ref[i][0], outs[i][0]) and torch.equal(ref[i][1], outs[i][1]), i
        with pytest.raises(eng.StereoError):
            e.wait_batches(len(ins) + 1)  # more than were submitted since the last wait()
        e.wait()
        # host-memory batches: complete = the maps are in the caller's arrays
        h_out = [(np.full((B, H, W), -77.0, np.float32), np.full((B, H, W), -77.0, np.float32)) for _ in batches]
        keep = [(np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])) for b in batches]
        for (l, r), (o1, o2) in zip(keep, h_out):
            e.submit_host(l, r, o1, o2)
        for i in range(len(batches)):
            e.wait_batches(i + 1)
            assert np.array_equal(h_out[i][0], ref[i][0].cpu().numpy()) and np.array_equal(h_out[i][1], ref[i][1].cpu().numpy()), i
        e.wait()
    finally:
        e.close()


def test_kernel_timing_selection(eng):
    """sv_kernel_timing_select: only the named kernels get HIP events; unknown names are refused."""
    import torch
    synth = util.pkg("synth")
    H, W, D, B = 96, 256, 48, 4
    batch = synth.make_batch(900, B, H, W, D)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=4, n_slots=2)
    try:
        left, right = torch.from_numpy(batch[:, 0].copy()).cuda(), torch.from_numpy(batch[:, 1].copy()).cuda()
        e.timing(True, only=("dense_match", "descriptor"))
        e.process_device(left, right)
        kt = {k: v for k, v in e.kernel_times().items() if not k.startswith("host:")}
        assert kt["dense_match"][1] > 0 and kt["descriptor"][1] > 0
        assert all(v[1] == 0 for k, v in kt.items() if k not in ("dense_match", "descriptor")), kt
        with pytest.raises(ValueError):
            e.timing(True, only=("no_such_kernel",))
        e.timing(False)
    finally:
        e.close()



def test_handle_on_second_device(eng, oracle):
    """A handle on device 1 (bench.py --gpus N gives every rank its own ordinal): control threads, pool threads and the
    lattice-filter fetch path all have to address that device.  Skipped on one-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 6
    batch = synth.make_batch(640, B, H, W, D)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), device=1, chunk=4, n_slots=2, n_workers=3)
    try:
        with torch.cuda.device(1):
            left, right = torch.from_numpy(batch[:, 0].copy()).to("cuda:1"), torch.from_numpy(batch[:, 1].copy()).to("cuda:1")
            d1, d2 = e.process_device(left, right)
            torch.cuda.synchronize()
        h1, h2, st = e.process_host(batch[:, 0], batch[:, 1])
        d1, d2 = d1.cpu().numpy(), d2.cpu().numpy()
    finally:
        e.close()
    assert np.array_equal(h1, d1) and np.array_equal(h2, d2) and (st >= 3).all()
    po = ElasParams.driver(D - 1)
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i


def _lattice_set(rng, W, H, step, D, n_lat, side, small_d=False, zero_corner=False):
    """Support-point-like vertex set: n_lat lattice points in scan order (u outer, v inner) with random disparities + the six corner
    points (elas.cpp:235-264), as the (x, y) the triangulation of `side` sees (elas.cpp:449-461); also returns the disparities."""
    Wc, Hc = -(-W // step), -(-H // step)
    cells = np.sort(rng.choice((Wc - 1) * (Hc - 1), n_lat, replace=False))
    u, v = (cells // (Hc - 1) + 1) * step, (cells % (Hc - 1) + 1) * step
    d = rng.integers(0, 4 if small_d else D + 1, n_lat)
    cd = rng.integers(1, D + 1, 4)
    if zero_corner:
        cd[2] = 0  # the top-right corner's nearest support point has disparity 0: (W-1, 0, 0) twice, in both images
    U = np.concatenate([u, [0, 0, W - 1, W - 1, W - 1 + cd[2], W - 1 + cd[3]]])
    V = np.concatenate([v, [0, H - 1, 0, H - 1, 0, H - 1]])
    Dd = np.concatenate([d, [cd[0], cd[1], cd[2], cd[3], cd[2], cd[3]]])
    return np.stack([U - Dd if side else U, V], 1).astype(np.int32), Dd.astype(np.int32)


def test_gpu_vertex_preparation_matches_host(eng):
    """Sort, duplicate scan and k-d order of a vertex set on the GPU (delaunay_gpu.hip: dg_prepare - ranks from bit maps of the
    occupied lattice cells, the alternating cuts one tree depth per pass) against Delaunay::prepare of the host stage, which follows
    the reference (triangle.cpp:5183-5360, 5889-5903).  Without coincident points: exactly the host's order.  Coincident points with
    equal disparities (the corner point with disparity 0, 7 of the 21 kitti_mini frames) are the same support point twice: the
    kernel keeps the lowest id, the host whichever the reference's quicksort puts first - the same vertices in the same order up
    to that label.  Coincident points with different disparities are not interchangeable: the kernel must hand the set back."""
    rng = np.random.default_rng(3)
    handed_back = merged = 0
    for it in range(150):
        W, H, step, D = [(1242, 375, 5, 127), (320, 120, 5, 63), (1242, 375, 5, 255), (640, 480, 3, 100), (203, 97, 5, 31), (1242, 375, 6, 127)][it % 6]
        Wc, Hc = -(-W // step), -(-H // step)
        n_lat = int(rng.integers(3, min(3700, (Wc - 1) * (Hc - 1))))
        side = it % 2
        xy, dd = _lattice_set(rng, W, H, step, D, n_lat, side=side, small_d=(it % 4 == 0) or (it % 5 == 0), zero_corner=(it % 5 == 0))
        if it % 5 == 0 and side:  # keep the right image free of other coincidences: one disparity everywhere except the corners
            dd[:-6] = 2
            xy, _ = _lattice_set(np.random.default_rng(it), W, H, step, D, n_lat, side=0)
            U = xy[:, 0].copy()
            U[-2:] = xy[-4:-2, 0] + dd[-2:]
            xy = np.stack([U - dd, xy[:, 1]], 1).astype(np.int32)
        want, got = eng.host_kd_order(xy), eng.gpu_kd_order(xy, W, H, step, D, disp=dd)
        keys = xy[:, 0].astype(np.int64) * 100000 + xy[:, 1]
        interchangeable = all(len(set(dd[keys == k])) == 1 for k in np.unique(keys[np.isin(keys, keys[np.unique(keys, return_counts=True, return_index=True)[1]])]) if (keys == k).sum() > 1)
        unique = len(np.unique(keys)) == len(keys)
        if got is None:
            handed_back += 1
            assert not unique and not interchangeable, "case %d: handed back although every coincident group is interchangeable" % it
            continue
        assert unique or interchangeable, "case %d: coincident points with different disparities not noticed" % it
        assert len(got) == len(want) == len(np.unique(keys))
        if unique:
            assert np.array_equal(got, want), "case %d: order differs" % it
        else:
            merged += 1
            assert np.array_equal(xy[got], xy[want]) and np.array_equal(dd[got], dd[want]), "case %d: order differs" % it
            assert all(got[i] == min(np.nonzero(keys == keys[got[i]])[0]) for i in range(len(got))), "case %d: not the lowest id of a coincident group" % it
    assert handed_back > 0 and merged > 0


@pytest.mark.parametrize("name", ["kitti0_d128", "kitti3_d128", "kitti7_d128", "kitti10_d128", "kitti0_d256_sub", "cones_crop_robotics", "synth7_d64"])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_resident_and_round3_triangulation_paths_agree(eng, oracle, monkeypatch, name, resident):
    """All-GPU triangulation with the support lists resident on the device (k_delaunay_resident: the host reads 8 meta words per pair)
    and with SV_RESIDENT=0 the round-3 path (lists to the host, vertex order from the pool, k_delaunay_blob): same maps, bit for bit."""
    entry = DIG[name]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=3, n_streams=2, n_workers=1, triangulation="gpu", resident=None if resident == "1" else False)
    try:
        assert e.query()["resident"] == int(resident)
        d1, d2, st = e.process_host(np.stack([L] * 9), np.stack([R] * 9))
    finally:
        e.close()
    assert (st == entry["n_support"]).all()
    for i in range(9):
        assert util.sha(d1[i]) == entry["stages"]["final1"] and util.sha(d2[i]) == entry["stages"]["final2"]


def test_resident_chunks_hand_coincident_points_to_the_host(eng, oracle, monkeypatch):
    """lr_threshold = 6 lets two support points of one lattice row match the same right-image column (|d1 - d2| = 5 or 10): coincident
    vertices in the right image with DIFFERENT disparities, whose survivor the reference's randomised quicksort decides.  The resident
    kernel must hand such a side to the host stage (counted as fallbacks) and the maps must still equal the oracle's."""
    synth = util.pkg("synth")
    H, W, D, B = 120, 320, 64, 13
    batch = synth.make_batch(900, B, H, W, D)
    p, po = eng.SvParams.driver(D - 1), ElasParams.driver(D - 1)
    for q in (p, po):
        q.lr_threshold, q.support_threshold, q.incon_min_support, q.incon_threshold = 6, 1.0, 1, 30
    e = eng.StereoEngine(W, H, p, chunk=4, n_slots=3, n_streams=2, n_workers=2, triangulation="gpu")
    try:
        assert e.query()["resident"] == 1
        d1, d2, st = e.process_host(batch[:, 0], batch[:, 1])
        fallbacks = e.gpu_triangulation_fallbacks()
    finally:
        e.close()
    for i in range(B):
        o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), i
    assert fallbacks > 0, "no pair of this batch had coincident points: the test does not exercise the hand-back"


@pytest.mark.parametrize("name", ["kitti13_d128", "kitti17_d128", "kitti20_d128"])
def test_resident_chunks_keep_the_zero_disparity_corner_on_the_device(eng, oracle, monkeypatch, name):
    """kitti_mini frames 13, 17, 20: the top-right image corner takes disparity 0, so elas.cpp:258-259 adds the support point (W-1, 0, 0) a
    second time - coincident vertices in both images, one third of the kitti_mini frames.  They are interchangeable (same triple), so
    the resident kernel drops one itself: no side is handed to the host, and the maps are the reference's."""
    entry = DIG[name]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=1)  # (one host thread: the GPU triangulates by itself)
    try:
        d1, d2, st = e.process_host(np.stack([L] * 5), np.stack([R] * 5))
        assert e.query()["resident"] == 1 and e.gpu_triangulation_fallbacks() == 0
    finally:
        e.close()
    assert (st == entry["n_support"]).all()
    for i in range(5):
        assert util.sha(d1[i]) == entry["stages"]["final1"] and util.sha(d2[i]) == entry["stages"]["final2"]


def test_resident_launch_with_too_little_lds_hands_sets_to_the_host(eng, oracle):
    """The resident kernel's LDS request follows the support counts of earlier chunks; a chunk whose sets are larger than the
    request covers is handed to the host stage side by side (and the request grows).  Forced here with the "ns_bound" hook."""
    entry = DIG["kitti0_d128"]
    L, R = util.case_images(entry)
    e = eng.StereoEngine(L.shape[1], L.shape[0], util.case_params(entry, eng.SvParams), chunk=4, n_slots=2, n_streams=2, n_workers=2, triangulation="gpu")
    try:
        e.debug_set("ns_bound", 500)  # kitti0 has 2 094 support points
        d1, d2, st = e.process_host(np.stack([L] * 4), np.stack([R] * 4))
        first = e.gpu_triangulation_fallbacks()
        d1b, d2b, _ = e.process_host(np.stack([L] * 4), np.stack([R] * 4))
        later = e.gpu_triangulation_fallbacks() - first
    finally:
        e.close()
    assert first == 8 and later == 0  # both sides of the four pairs of the first chunk; the bound has grown since
    for a, b in ((d1, d2), (d1b, d2b)):
        for i in range(4):
            assert util.sha(a[i]) == entry["stages"]["final1"] and util.sha(b[i]) == entry["stages"]["final2"]


def test_policy_fields_of_the_configuration(eng):
    """sv_config's policy fields decide what the environment variables used to: two handles of one process configured differently."""
    p = eng.SvParams.driver(63)
    a = eng.StereoEngine(320, 120, p, chunk=4, n_slots=2, n_workers=3, triangulation="host", gpu_filter=False, affinity=False)
    b = eng.StereoEngine(320, 120, p, chunk=4, n_slots=2, n_workers=3, triangulation="gpu", gpu_filter=True)
    try:
        qa, qb = a.query(), b.query()
    finally:
        a.close()
        b.close()
    assert (qa["gpu_lattice_filter"], qa["gpu_triangulation"], qa["resident"], qa["numa_bound"]) == (0, 0, 0, 0)
    assert (qb["gpu_lattice_filter"], qb["gpu_triangulation"], qb["resident"]) == (1, 1, 1)


@pytest.mark.parametrize("mode", ["block", "spin", "poll"])
def test_the_ways_host_threads_wait_give_the_same_maps(eng, mode):
    """sv_config.event_sync: hipEventBlockingSync events, spinning, or asking the event between 40 us naps (the default for chunks of
    four pairs and more) - streamed batches through a small pipeline, device and host memory, against the automatic mode."""
    import torch
    B, H, W, D = 12, 120, 320, 64
    batch = util.pkg("synth").make_batch(4200, B, H, W, D)
    left = torch.from_numpy(np.ascontiguousarray(batch[:, 0])).cuda()
    right = torch.from_numpy(np.ascontiguousarray(batch[:, 1])).cuda()
    p = eng.SvParams.driver(D - 1)
    ref = eng.StereoEngine(W, H, p, chunk=4, n_slots=2)
    e = eng.StereoEngine(W, H, p, chunk=4, n_slots=2, event_sync=mode)
    try:
        r1, r2 = ref.process_device(left, right)
        outs = [(torch.empty_like(r1), torch.empty_like(r2)) for _ in range(3)]
        for d1, d2 in outs:  # three batches in flight
            e.submit_device(left, right, d1, d2)
        e.wait()
        for d1, d2 in outs:
            assert torch.equal(d1, r1) and torch.equal(d2, r2)
        h1, h2, _ = e.process_host(np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1]), want_d2=True)
        assert np.array_equal(h1, r1.cpu().numpy()) and np.array_equal(h2, r2.cpu().numpy())
    finally:
        e.close()
        ref.close()


@pytest.mark.parametrize("hook", ["lat_runtime_copies", "lat_filter_alone", "latency_pin_off"])
def test_single_pair_paths_behind_the_test_hooks_give_the_same_maps(eng, hook):
    """The single-pair path has alternatives the round measured against each other and kept reachable: the lattice / blob copies through
    hipMemcpyAsync instead of the copy kernel, the lattice filters on the calling thread alone instead of as a team, the polling threads
    left where the pool runs.  Same maps either way, device and page-locked host memory."""
    import torch
    l, r = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
    H, W = l.shape
    p = eng.SvParams.driver(127)
    ref = eng.StereoEngine(W, H, p, chunk=4, n_slots=2)
    e = eng.StereoEngine(W, H, p, chunk=1, n_slots=2, n_streams=1, n_workers=7)
    try:
        L, R = torch.from_numpy(np.ascontiguousarray(l[None])).cuda(), torch.from_numpy(np.ascontiguousarray(r[None])).cuda()
        r1, r2 = ref.process_device(L, R)
        e.process_device(L, R)
        e.debug_set("latency_pin" if hook == "latency_pin_off" else hook, 0 if hook == "latency_pin_off" else 1)
        for _ in range(3):
            d1, d2 = e.process_device(L, R)
            assert torch.equal(d1, r1) and torch.equal(d2, r2)
        hl, hr = eng.pinned_array((H, W), np.uint8), eng.pinned_array((H, W), np.uint8)
        h1, h2 = eng.pinned_array((H, W), np.float32), eng.pinned_array((H, W), np.float32)
        hl[:], hr[:] = l, r
        dims = (ctypes.c_int32 * 3)(W, H, W)
        assert eng.lib().sv_elas_process(e._h, hl.ctypes.data, hr.ctypes.data, h1.ctypes.data, h2.ctypes.data, dims) == 0
        assert np.array_equal(h1, r1[0].cpu().numpy()) and np.array_equal(h2, r2[0].cpu().numpy())
    finally:
        e.close()
        ref.close()


@pytest.mark.parametrize("split", [0, 1, 2, 3])
def test_single_pairs_with_shared_triangulations_give_the_same_maps(eng, split):
    """Latency mode (one pair per call on a chunk-1 handle): each triangulation on one thread (3), or its top-level cuts shared with pool
    threads (sv_config.latency_split = 1: halves, 2: quarters; 0: halves when the helpers can sit on the caller's L3, the default) - same maps
    as a throughput handle; the real frame and a synthetic one."""
    import torch
    l, r = util.load_png("kitti0_left.png"), util.load_png("kitti0_right.png")
    H, W = l.shape
    p = eng.SvParams.driver(127)
    syn = util.pkg("synth").make_batch(77, 1, H, W, 128)[0]
    ref = eng.StereoEngine(W, H, p, chunk=4, n_slots=2)
    e = eng.StereoEngine(W, H, p, chunk=1, n_slots=2, n_streams=1, n_workers=8, latency_split=split)
    try:
        assert e.query()["latency_split"] == ({1: 1, 2: 2, 3: 0}[split] if split else e.query()["latency_split"]) and e.query()["latency_split"] in (0, 1, 2)
        for a, b in ((l, r), (syn[0], syn[1])):
            L, R = torch.from_numpy(np.ascontiguousarray(a[None])).cuda(), torch.from_numpy(np.ascontiguousarray(b[None])).cuda()
            r1, r2 = ref.process_device(L, R)
            for _ in range(3):  # (repeated: the pool's pollers pick the pieces up in varying order)
                d1, d2 = e.process_device(L, R)
                assert torch.equal(d1, r1) and torch.equal(d2, r2)
    finally:
        e.close()
        ref.close()
