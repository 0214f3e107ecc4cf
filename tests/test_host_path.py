"""GPU (MI355X): the streamed host-memory path (sv_process_batch_host / sv_submit_batch_host / sv_elas_process) - the form of
the reference's seam, which takes host pointers (elas.h:162, stereo_vision.cpp:313).  Pageable and page-locked caller
memory, several chunks in flight, padded rows, pairs the reference leaves untouched; every map bit-exact vs the oracle."""
import ctypes

import numpy as np
import pytest

import util
from pyoracle import ElasParams

pytestmark = pytest.mark.gpu

H, W, D = 120, 320, 64


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no CPU fallback"
    return util.pkg("engine")


@pytest.fixture(scope="module")
def batch13(oracle):
    synth = util.pkg("synth")
    B = 13
    batch = synth.make_batch(700, B, H, W, D)
    po = ElasParams.driver(D - 1)
    want = [oracle.process(po, batch[i, 0], batch[i, 1])[:2] for i in range(B)]
    return batch, want


def _check(d1, d2, want):
    for i, (o1, o2) in enumerate(want):
        assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)), "pair %d D1" % i
        if d2 is not None:
            assert np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), "pair %d D2" % i


@pytest.mark.parametrize("copies", ["lanes", "runtime"])
@pytest.mark.parametrize("memory", ["pageable", "pinned", "pinned_forced_staging"])
def test_host_batch_streams_through_several_chunks(eng, batch13, memory, copies, monkeypatch):
    """13 pairs, chunk 2, 4 slots: >= 3 chunks in flight, the last chunk is ragged; twice on the same handle.  copies: who moves the
    chunks over PCIe - SDMA engines the engine addresses itself (csrc/dma_lanes.cpp; what a handle picks by default on this runtime) or
    hipMemcpyAsync on the copy streams (sv_config.host_copies)."""
    batch, want = batch13
    B = batch.shape[0]
    if memory == "pageable":
        L, R = np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1])
        d1, d2 = np.full((B, H, W), 7.0, np.float32), np.full((B, H, W), 7.0, np.float32)
    else:
        L, R = eng.pinned_array((B, H, W), np.uint8), eng.pinned_array((B, H, W), np.uint8)
        L[:], R[:] = batch[:, 0], batch[:, 1]
        d1, d2 = eng.pinned_array((B, H, W), np.float32), eng.pinned_array((B, H, W), np.float32)
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=3, chunk=2, n_streams=2, n_slots=4, host_copies=copies)
    try:
        assert e.query()["host_copies"] == 0  # decided at the first host-memory batch
        if memory == "pinned_forced_staging":
            e.debug_set("host_force_staging", 1)
        for _ in range(2):
            d1[:], d2[:] = 7.0, 7.0
            o1, o2, status = e.process_host(L, R, d1=d1, d2=d2)
            assert o1 is d1 and o2 is d2 and (status >= 3).all()
            _check(d1, d2, want)
            assert e.query()["host_copies"] == (2 if copies == "lanes" else 1)
        d1[:] = 7.0
        e.process_host(L, R, want_d2=False, d1=d1)  # D2 == NULL
        _check(d1, None, want)
    finally:
        e.close()


def test_host_batch_padded_rows_and_untouched_pairs(eng, oracle):
    """stride > width (dims[2] of Elas::process); ROBOTICS preset with a textureless pair in the middle of a chunk: fewer than 3
    support points -> the caller's maps of that pair keep their bytes (elas.cpp:63-69), its neighbours are processed."""
    synth = util.pkg("synth")
    B, stride = 5, W + 24
    batch = synth.make_batch(820, B, H, W, D)
    batch[2] = 0
    p = eng.SvParams.preset("robotics")
    p.disp_max = D - 1
    po = ElasParams.preset("robotics")
    po.disp_max = D - 1
    for pinned in (False, True):
        alloc = eng.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
        L, R = alloc((B, H, stride), np.uint8), alloc((B, H, stride), np.uint8)
        L[:], R[:] = 99, 99  # the padding must never be read as image
        L[:, :, :W], R[:, :, :W] = batch[:, 0], batch[:, 1]
        d1, d2 = alloc((B, H, W), np.float32), alloc((B, H, W), np.float32)
        d1[:], d2[:] = 3.5, 4.5
        status = np.zeros(B, np.int32)
        e = eng.StereoEngine(W, H, p, n_workers=2, chunk=4, n_slots=2)
        try:
            rc = eng.lib().sv_process_batch_host(e._h, L.ctypes.data, R.ctypes.data, B, stride, d1.ctypes.data, d2.ctypes.data, status.ctypes.data)
            assert rc == 0, eng.lib().sv_last_error(e._h)
        finally:
            e.close()
        assert status[2] < 3 and (d1[2] == 3.5).all() and (d2[2] == 4.5).all()
        for i in (0, 1, 3, 4):
            o1, o2, _ = oracle.process(po, batch[i, 0], batch[i, 1])
            assert status[i] >= 3
            assert np.array_equal(d1[i].view(np.uint8), o1.view(np.uint8)) and np.array_equal(d2[i].view(np.uint8), o2.view(np.uint8)), (pinned, i)


def test_host_streamed_submissions(eng, batch13):
    """sv_submit_batch_host x3 + sv_wait: batches follow each other through the ring; mixed with a device-memory batch."""
    import torch
    batch, want = batch13
    B = batch.shape[0]
    L, R = np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1])
    outs = [(np.zeros((B, H, W), np.float32), np.zeros((B, H, W), np.float32)) for _ in range(3)]
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=4, chunk=4, n_slots=3)
    try:
        dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        t1, t2 = torch.zeros((B, H, W), device="cuda"), torch.zeros((B, H, W), device="cuda")
        torch.cuda.synchronize()
        e.submit_host(L, R, *outs[0])
        e.submit_device(dl, dr, t1, t2)
        e.submit_host(L, R, *outs[1])
        e.submit_host(L, R, *outs[2])
        e.wait()
        torch.cuda.synchronize()
        for o1, o2 in outs:
            _check(o1, o2, want)
        _check(t1.cpu().numpy(), t2.cpu().numpy(), want)
    finally:
        e.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_elas_process_seam_latency_mode(eng, pinned):
    """Elas::process's own signature on a chunk = 1 handle: the calling thread drives the pair through persistent device
    buffers (no allocation per call); repeated calls, golden digest of the reference."""
    entry = util.digests()["kitti0_crop_d64"]
    L0, R0 = util.case_images(entry)
    h, w = L0.shape
    alloc = eng.pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
    L, R, D1, D2 = alloc((h, w), np.uint8), alloc((h, w), np.uint8), alloc((h, w), np.float32), alloc((h, w), np.float32)
    L[:], R[:] = L0, R0
    dims = (ctypes.c_int32 * 3)(w, h, w)
    e = eng.StereoEngine(w, h, util.case_params(entry, eng.SvParams), n_workers=4, chunk=1, n_streams=1, n_slots=2)
    try:
        for _ in range(3):
            D1[:], D2[:] = 0, 0
            rc = eng.lib().sv_elas_process(e._h, L.ctypes.data, R.ctypes.data, D1.ctypes.data, D2.ctypes.data, dims)
            assert rc == 0, eng.lib().sv_last_error(e._h)
            assert util.sha(D1) == entry["stages"]["final1"] and util.sha(D2) == entry["stages"]["final2"]
    finally:
        e.close()


def test_device_batch_arguments_are_validated(eng):
    """submit_device / process_device refuse what the kernels cannot take (they work on raw pointers)."""
    import torch
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=2, n_slots=2)
    try:
        l = torch.zeros((2, H, W), dtype=torch.uint8, device="cuda")
        d = torch.zeros((2, H, W), dtype=torch.float32, device="cuda")
        with pytest.raises(ValueError):
            e.submit_device(l, l[:, :, : W - 8], d)                      # shape / contiguity
        with pytest.raises(ValueError):
            e.submit_device(l, l, d[:, : H - 1])                          # map size
        with pytest.raises(ValueError):
            e.submit_device(l, l.to(torch.int8), d)                       # dtype
        with pytest.raises(ValueError):
            e.submit_device(l.cpu(), l, d)                                # host tensor
        with pytest.raises(ValueError):
            e.process_device(torch.zeros((2, H + 2, W), dtype=torch.uint8, device="cuda"), l)
        e.submit_device(l, l, d, d.clone())
        e.wait()
    finally:
        e.close()


@pytest.mark.parametrize("memory", ["pinned", "pageable"])
def test_host_batch_with_the_drivers_8bit_output(eng, batch13, memory):
    """sv_process_batch_host_dmap / sv_submit_batch_host_dmap: the driver's output format (dmap = saturate(round_half_even(4 * D1)),
    stereo_vision.cpp:316) straight from the device - equals the conversion of the float maps of the same handle; several chunks,
    a ragged last one, streamed submissions, mixed with float jobs; an image of a pair with < 3 support points stays untouched."""
    batch, _want = batch13
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), chunk=4, n_slots=3, n_workers=3)
    alloc = eng.pinned_array if memory == "pinned" else (lambda shape, dt: np.zeros(shape, dt))
    try:
        L, R = alloc(batch[:, 0].shape, np.uint8), alloc(batch[:, 0].shape, np.uint8)
        L[:], R[:] = batch[:, 0], batch[:, 1]
        L[5] = 0
        R[5] = 0  # textureless: only the six corner points with the driver preset -> still >= 3; see the robotics part below
        d1, _, st = e.process_host(L, R, want_d2=False)
        want = np.clip(np.rint(d1 * np.float32(4.0)), 0, 255).astype(np.uint8)
        dm = alloc(want.shape, np.uint8)
        dm[:] = 77
        got, st2 = e.process_host_dmap(L, R, dmap=dm)
        assert np.array_equal(st, st2) and np.array_equal(got, want)
        a, b = alloc(want.shape, np.uint8), alloc(want.shape, np.uint8)
        f1 = alloc(d1.shape, np.float32)
        e.submit_host_dmap(L, R, a)
        e.submit_host(L, R, f1)
        e.submit_host_dmap(L, R, b)
        e.wait()
        assert np.array_equal(a, want) and np.array_equal(b, want) and np.array_equal(f1, d1)
    finally:
        e.close()
    p = eng.SvParams.preset("robotics")
    p.disp_max = D - 1
    e = eng.StereoEngine(W, H, p, chunk=4, n_slots=2, n_workers=2)
    try:
        Z = np.zeros_like(batch[:3, 0])
        Z[1] = batch[1, 0]
        ZR = np.zeros_like(Z)
        ZR[1] = batch[1, 1]
        dm = np.full(Z.shape, 99, np.uint8)
        got, st = e.process_host_dmap(Z, ZR, dmap=dm)
        assert st[0] < 3 and st[2] < 3 and st[1] >= 3
        assert (got[0] == 99).all() and (got[2] == 99).all() and not (got[1] == 99).all()  # elas.cpp:63-69: untouched outputs
    finally:
        e.close()


def test_dma_lanes_that_fail_their_self_test_fall_back_to_the_runtime(eng, batch13):
    """Before a batch depends on them the SDMA lanes move a few bytes both ways; when that fails (forced here) a handle with the automatic
    policy uses hipMemcpyAsync instead - same maps - and a handle that was told to use lanes reports it instead of falling back silently."""
    batch, want = batch13
    L, R = np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1])
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=3, chunk=4, n_slots=3)
    try:
        e.debug_set("dma_selftest_fail", 1)
        d1, d2, status = e.process_host(L, R)
        assert e.query()["host_copies"] == 1 and (status >= 3).all()
        _check(d1, d2, want)
    finally:
        e.close()
    e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=3, chunk=4, n_slots=3, host_copies="lanes")
    try:
        e.debug_set("dma_selftest_fail", 1)
        with pytest.raises(eng.StereoError, match="self-test"):
            e.process_host(L, R)
    finally:
        e.close()


def test_registered_host_memory_is_page_locked_memory(eng, batch13):
    """Caller memory that was page-locked after the fact (hipHostRegister - what an OpenCV / numpy user has) takes the same paths as
    hipHostMalloc'ed memory: DMA lanes for a streamed batch, no copies at all for single pairs on a latency handle (the kernels use the
    device-side address of the registration)."""
    import torch
    rt = torch.cuda.cudart()
    batch, want = batch13
    B = batch.shape[0]
    bufs = [np.ascontiguousarray(batch[:, 0]), np.ascontiguousarray(batch[:, 1]), np.full((B, H, W), 5.0, np.float32), np.full((B, H, W), 5.0, np.float32)]
    for a in bufs:
        assert int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) == 0
    try:
        L, R, d1, d2 = bufs
        e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=3, chunk=4, n_slots=3)
        try:
            e.process_host(L, R, d1=d1, d2=d2)
            assert e.query()["host_copies"] == 2
            _check(d1, d2, want)
        finally:
            e.close()
        d1[:], d2[:] = 5.0, 5.0
        e = eng.StereoEngine(W, H, eng.SvParams.driver(D - 1), n_workers=2, chunk=1, n_streams=1, n_slots=2)
        try:
            for i in range(3):
                o1, o2, st = e.process_host(L[i], R[i], d1=d1[i:i + 1], d2=d2[i:i + 1])
            _check(d1[:3], d2[:3], want[:3])
        finally:
            e.close()
    finally:
        for a in bufs:
            rt.cudaHostUnregister(a.ctypes.data)
