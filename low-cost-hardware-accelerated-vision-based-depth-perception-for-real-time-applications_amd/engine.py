"""Python front-end of libstereo_vision_hip.so (ctypes over the C ABI of include/stereo_vision_hip.h).

PyTorch is used only as plumbing: device memory (tensors), the current device and torch.distributed in bench.py.
All compute is in the hand-written HIP kernels behind the C ABI; there is no eager/CPU fallback — if the
library is missing or no GPU is present, calls fail loudly.
"""
import ctypes
import os

import numpy as np

# The engine drives six HIP streams (two for the first GPU phase, four for the second) next to the application's own; ROCm
# maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.  8 measured +2 %
# pairs/s over the default, 6 slightly less than the default.  Only effective before the process's first HIP call;
# an explicit setting of the application wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SV_LIB_PATH") or os.path.join(HERE, "libstereo_vision_hip.so")  # SV_LIB_PATH: kernel experiments only

SV_ROBOTICS, SV_MIDDLEBURY, SV_DRIVER = 0, 1, 2


class SvParams(ctypes.Structure):
    """sv_params == Elas::parameters (reference: src/serial_includes/elas/elas.h:60-145)."""

    _fields_ = [
        ("disp_min", ctypes.c_int32), ("disp_max", ctypes.c_int32), ("support_threshold", ctypes.c_float),
        ("support_texture", ctypes.c_int32), ("candidate_stepsize", ctypes.c_int32), ("incon_window_size", ctypes.c_int32),
        ("incon_threshold", ctypes.c_int32), ("incon_min_support", ctypes.c_int32), ("add_corners", ctypes.c_int32),
        ("grid_size", ctypes.c_int32), ("beta", ctypes.c_float), ("gamma", ctypes.c_float), ("sigma", ctypes.c_float),
        ("sradius", ctypes.c_float), ("match_texture", ctypes.c_int32), ("lr_threshold", ctypes.c_int32),
        ("speckle_sim_threshold", ctypes.c_float), ("speckle_size", ctypes.c_int32), ("ipol_gap_width", ctypes.c_int32),
        ("filter_median", ctypes.c_int32), ("filter_adaptive_mean", ctypes.c_int32), ("postprocess_only_left", ctypes.c_int32),
        ("subsampling", ctypes.c_int32),
    ]

    @classmethod
    def preset(cls, setting):
        p = cls()
        code = {"robotics": SV_ROBOTICS, "middlebury": SV_MIDDLEBURY, "driver": SV_DRIVER}[setting]
        lib().sv_params_init(ctypes.byref(p), code)
        return p

    @classmethod
    def driver(cls, disp_max=255):
        """MIDDLEBURY + postprocess_only_left + adaptive mean: what the reference driver runs (stereo_vision.cpp:307-311)."""
        p = cls.preset("driver")
        p.disp_max = disp_max
        return p


class SvConfig(ctypes.Structure):
    """sv_config of include/stereo_vision_hip.h: 0 = the default for every field."""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("device", ctypes.c_int32), ("n_workers", ctypes.c_int32),
                ("chunk", ctypes.c_int32), ("keep_debug", ctypes.c_int32), ("n_streams", ctypes.c_int32), ("n_slots", ctypes.c_int32),
                ("gpu_lattice_filter", ctypes.c_int32), ("gpu_triangulation", ctypes.c_int32), ("gpu_triangulation_pct", ctypes.c_int32),
                ("resident", ctypes.c_int32), ("dg_sub_max", ctypes.c_int32), ("dg_max_points", ctypes.c_int32), ("affinity", ctypes.c_int32),
                ("inline_latency_path", ctypes.c_int32), ("event_sync", ctypes.c_int32), ("share_sliced", ctypes.c_int32), ("latency_split", ctypes.c_int32), ("host_copies", ctypes.c_int32), ("reserved", ctypes.c_int32 * 4)]


_TRIANGULATION_MODES = {None: 0, "auto": 0, "gpu": 1, "host": 2, "balanced": 4}


_lib = None

_STAGE_DTYPES = {"desc1": np.uint8, "desc2": np.uint8, "dcan_raw": np.int16, "dcan_dims": np.int32, "support": np.int32,
                 "tri1": np.int32, "tri2": np.int32, "grid1": np.int32, "grid2": np.int32, "grid_dims": np.int32,
                 "tri_id1": np.int32, "tri_id2": np.int32}


def share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64; the library links the system one.  Whichever is loaded first serves both
    (same SONAME), and only PyTorch's copy works for PyTorch: import torch first whenever it is installed, so that tensors
    and the engine share one HIP runtime no matter in which order the application touches them."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """Loads the shared library (building it in-tree with hipcc if it is not there yet)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    share_hip_runtime_with_torch()
    L = ctypes.CDLL(LIB_PATH)
    u8p, f32p, i32p = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
    L.sv_params_init.argtypes = [ctypes.POINTER(SvParams), ctypes.c_int]
    L.sv_params_init.restype = None
    L.sv_create.argtypes = [ctypes.POINTER(SvParams), ctypes.POINTER(SvConfig), ctypes.POINTER(ctypes.c_void_p)]
    L.sv_create.restype = ctypes.c_int
    L.sv_destroy.argtypes = [ctypes.c_void_p]
    L.sv_destroy.restype = ctypes.c_int
    L.sv_last_error.argtypes = [ctypes.c_void_p]
    L.sv_last_error.restype = ctypes.c_char_p
    L.sv_wait.argtypes = [ctypes.c_void_p]
    L.sv_wait.restype = ctypes.c_int
    L.sv_wait_batches.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sv_wait_batches.restype = ctypes.c_int
    L.sv_host_alloc.argtypes = [ctypes.c_size_t]
    L.sv_host_alloc.restype = ctypes.c_void_p
    L.sv_host_free.argtypes = [ctypes.c_void_p]
    L.sv_host_free.restype = None
    L.sv_query.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sv_query.restype = ctypes.c_int
    for name in ("sv_process_batch_host_dmap", "sv_submit_batch_host_dmap"):
        f = getattr(L, name)
        f.argtypes = [ctypes.c_void_p, u8p, u8p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, i32p]
        f.restype = ctypes.c_int
    for name in ("sv_process_batch_device", "sv_process_batch_host", "sv_submit_batch_device", "sv_submit_batch_host"):
        f = getattr(L, name)
        f.argtypes = [ctypes.c_void_p, u8p, u8p, ctypes.c_int, ctypes.c_int, f32p, f32p, i32p]
        f.restype = ctypes.c_int
    L.sv_elas_process.argtypes = [ctypes.c_void_p, u8p, u8p, f32p, f32p, ctypes.POINTER(ctypes.c_int32)]
    L.sv_elas_process.restype = ctypes.c_int
    L.sv_debug_size.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.sv_debug_size.restype = ctypes.c_long
    L.sv_debug_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
    L.sv_debug_get.restype = ctypes.c_long
    L.sv_kernel_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    L.sv_kernel_times.restype = ctypes.c_int
    L.sv_kernel_times_reset.argtypes = [ctypes.c_void_p]
    L.sv_kernel_times_reset.restype = None
    L.sv_kernel_timing_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sv_kernel_timing_enable.restype = None
    L.sv_kernel_timing_select.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.sv_kernel_timing_select.restype = ctypes.c_int
    L.sv_host_support_filter.argtypes = [ctypes.POINTER(SvParams), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.sv_host_support_filter.restype = ctypes.c_int
    L.sv_host_support_filter_threads.argtypes = [ctypes.POINTER(SvParams), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.sv_host_support_filter_threads.restype = ctypes.c_int
    L.sv_host_delaunay.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.sv_host_delaunay.restype = ctypes.c_int
    _lib = L
    return L


class StereoError(RuntimeError):
    pass


class StereoEngine:
    """Batched Elas::process on one MI355X.

    engine = StereoEngine(1242, 375, SvParams.driver(127))
    d1, d2 = engine.process_device(left_u8_cuda, right_u8_cuda)     # torch tensors [B,H,W] -> float32 [B,H,W]
    """

    def __init__(self, width, height, params=None, device=0, n_workers=0, chunk=0, keep_debug=False, n_streams=0, n_slots=0, gpu_filter=None,
                 triangulation=None, resident=None, dg_sub_max=0, dg_max_points=0, affinity=None, inline=None, share_sliced=False, event_sync=None, latency_split=0, host_copies=None):
        """gpu_filter: None (automatic) / True / False - where the support-lattice filters run.  triangulation: None or "auto", "gpu", "host",
        "balanced" (by the pool's backlog, whatever its size) or an int 1..100 = that share of the chunks on the GPU.  resident / affinity /
        inline: None (automatic) or False to switch the resident GPU share / the NUMA binding / the calling-thread latency path off.
        dg_sub_max, dg_max_points: limits of the GPU triangulation (tests).  event_sync: None (automatic), "block", "spin" or "poll" - how the
        handle's threads wait for the GPU.  host_copies: None (automatic), "runtime" (hipMemcpyAsync) or "lanes" (engine-addressed SDMA copies) -
        who moves host-memory batches over PCIe.  See sv_config in include/stereo_vision_hip.h."""
        L = lib()
        self.params = params if params is not None else SvParams.driver(127)
        self.width, self.height, self.device = int(width), int(height), int(device)
        # disparity map size: half the image in half-resolution mode (Elas::parameters::subsampling, elas.h:83-85, 160-161)
        self.map_height, self.map_width = (self.height // 2, self.width // 2) if self.params.subsampling else (self.height, self.width)
        cfg = SvConfig(self.width, self.height, self.device, int(n_workers), int(chunk), int(bool(keep_debug)), int(n_streams), int(n_slots))
        cfg.gpu_lattice_filter = 0 if gpu_filter is None else (1 if gpu_filter else 2)
        if isinstance(triangulation, int) and not isinstance(triangulation, bool):
            cfg.gpu_triangulation, cfg.gpu_triangulation_pct = 3, int(triangulation)
        else:
            cfg.gpu_triangulation = _TRIANGULATION_MODES[triangulation]
        cfg.resident = 2 if resident is False else 0
        cfg.dg_sub_max, cfg.dg_max_points = int(dg_sub_max), int(dg_max_points)
        cfg.affinity = 2 if affinity is False else 0
        cfg.inline_latency_path = 2 if inline is False else 0
        cfg.share_sliced = int(bool(share_sliced))
        cfg.event_sync = {None: 0, "auto": 0, "block": 1, "spin": 2, "poll": 3}[event_sync]
        cfg.latency_split = int(latency_split)
        cfg.host_copies = {None: 0, "auto": 0, "runtime": 1, "lanes": 2}[host_copies]
        h = ctypes.c_void_p()
        rc = L.sv_create(ctypes.byref(self.params), ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise StereoError("sv_create failed (%d): %s" % (rc, L.sv_last_error(None).decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().sv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_set(self, key, value):
        """Test hooks on a live handle (sv_debug_set): "ccl_cap", "rt_cap", "host_force_staging", "ns_bound", "pool_sleep", "lat_trace", "dma_selftest_fail", "latency_pin", "lat_runtime_copies", "lat_filter_alone" (include/stereo_vision_hip.h)."""
        L = lib()
        L.sv_debug_set.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        self._check(L.sv_debug_set(self._h, key.encode(), int(value)))

    def _check(self, rc):
        if rc != 0:
            raise StereoError("libstereo_vision_hip error %d: %s" % (rc, lib().sv_last_error(self._h).decode()))

    # ---- device path (inputs resident in HBM)
    def _check_device_batch(self, left, right, d1, d2, status):
        """The engine reads and writes raw pointers on its own streams: everything the kernels assume is checked here, and work
        pending on torch's current stream (an H2D copy of the inputs, a fill of the outputs) is waited for."""
        import torch
        dev = torch.device("cuda", self.device)
        for t, dt, name in ((left, torch.uint8, "left"), (right, torch.uint8, "right"), (d1, torch.float32, "d1"), (d2, torch.float32, "d2")):
            if t is None and name == "d2":
                continue
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev):
                raise ValueError("%s must be a CUDA tensor on %s (the handle's device)" % (name, dev))
            if t.dtype != dt or not t.is_contiguous() or t.dim() != 3:
                raise ValueError("%s must be a contiguous %s tensor [B,H,W]" % (name, dt))
        B = left.shape[0]
        if tuple(left.shape) != (B, self.height, self.width) or right.shape != left.shape:
            raise ValueError("images must be [B,%d,%d], got %s / %s" % (self.height, self.width, tuple(left.shape), tuple(right.shape)))
        for t, name in ((d1, "d1"), (d2, "d2")):
            if t is not None and tuple(t.shape) != (B, self.map_height, self.map_width):
                raise ValueError("%s must be [%d,%d,%d], got %s" % (name, B, self.map_height, self.map_width, tuple(t.shape)))
        if status is not None and not (isinstance(status, np.ndarray) and status.dtype == np.int32 and status.flags.c_contiguous and status.size >= B):
            raise ValueError("status must be a contiguous int32 numpy array with at least B entries")
        torch.cuda.current_stream(dev).synchronize()
        return B

    def process_device(self, left, right, d1=None, d2=None, want_d2=True, status=None):
        import torch
        if isinstance(left, torch.Tensor) and isinstance(right, torch.Tensor):
            left, right = left.contiguous(), right.contiguous()
        B = left.shape[0]
        if d1 is None:
            d1 = torch.zeros((B, self.map_height, self.map_width), dtype=torch.float32, device=left.device)
        if d2 is None and want_d2:
            d2 = torch.zeros((B, self.map_height, self.map_width), dtype=torch.float32, device=left.device)
        self._check_device_batch(left, right, d1, d2, status)
        st = status.ctypes.data_as(ctypes.c_void_p) if status is not None else None
        self._check(lib().sv_process_batch_device(self._h, left.data_ptr(), right.data_ptr(), B, self.width, d1.data_ptr(),
                                                  d2.data_ptr() if d2 is not None else None, st))
        return d1, d2

    def submit_device(self, left, right, d1, d2=None, status=None):
        """Streaming form: enqueue a device-resident batch and return at once (call wait() before touching d1/d2).
        Successive batches flow through the pipeline back to back."""
        B = self._check_device_batch(left, right, d1, d2, status)
        st = status.ctypes.data_as(ctypes.c_void_p) if status is not None else None
        self._check(lib().sv_submit_batch_device(self._h, left.data_ptr(), right.data_ptr(), B, self.width, d1.data_ptr(),
                                                 d2.data_ptr() if d2 is not None else None, st))

    def wait(self):
        self._check(lib().sv_wait(self._h))

    def wait_batches(self, n):
        """Returns when the n oldest batches submitted since the last wait() are complete; later ones keep running."""
        self._check(lib().sv_wait_batches(self._h, int(n)))

    # ---- host path (numpy in / out, PCIe inclusive): streamed through the pipeline, no allocation per call
    def _host_args(self, left, right, d1, d2, want_d2):
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        if left.ndim == 2:
            left, right = left[None], right[None]
        B, H, W = left.shape
        if (H, W) != (self.height, self.width) or right.shape != left.shape:
            raise ValueError("images must be [B,%d,%d]" % (self.height, self.width))
        shape = (B, self.map_height, self.map_width)
        if d1 is None:
            d1 = np.zeros(shape, np.float32)
        if d2 is None and want_d2:
            d2 = np.zeros(shape, np.float32)
        for m, name in ((d1, "d1"), (d2, "d2")):
            if m is not None and not (isinstance(m, np.ndarray) and m.dtype == np.float32 and m.flags.c_contiguous and m.shape == shape):
                raise ValueError("%s must be a contiguous float32 array %s" % (name, shape))
        return left, right, d1, d2, B

    def process_host(self, left, right, want_d2=True, d1=None, d2=None):
        """numpy [B,H,W] uint8 in, float32 maps out.  Page-locked arrays (pinned_array) skip the staging copies."""
        left, right, d1, d2, B = self._host_args(left, right, d1, d2, want_d2)
        status = np.zeros(B, np.int32)
        self._check(lib().sv_process_batch_host(self._h, left.ctypes.data, right.ctypes.data, B, self.width, d1.ctypes.data,
                                                d2.ctypes.data if d2 is not None else None, status.ctypes.data))
        return d1, d2, status

    def submit_host(self, left, right, d1, d2=None, status=None):
        """Streaming form of process_host: returns at once; the arrays must stay alive and untouched until wait()."""
        left_c, right_c, d1, d2, B = self._host_args(left, right, d1, d2, False)
        if left_c is not left and not np.shares_memory(left_c, left) or right_c is not right and not np.shares_memory(right_c, right):
            raise ValueError("submit_host needs contiguous uint8 arrays (a temporary copy would be freed before the engine reads it)")
        st = status.ctypes.data_as(ctypes.c_void_p) if status is not None else None
        self._check(lib().sv_submit_batch_host(self._h, left_c.ctypes.data, right_c.ctypes.data, B, self.width, d1.ctypes.data,
                                               d2.ctypes.data if d2 is not None else None, st))

    def _host_dmap_args(self, left, right, dmap):
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        if left.ndim == 2:
            left, right = left[None], right[None]
        B, H, W = left.shape
        if (H, W) != (self.height, self.width) or right.shape != left.shape:
            raise ValueError("images must be [B,%d,%d]" % (self.height, self.width))
        shape = (B, self.map_height, self.map_width)
        if dmap is None:
            dmap = np.zeros(shape, np.uint8)
        if not (isinstance(dmap, np.ndarray) and dmap.dtype == np.uint8 and dmap.flags.c_contiguous and dmap.shape == shape):
            raise ValueError("dmap must be a contiguous uint8 array %s" % (shape,))
        return left, right, dmap, B

    def process_host_dmap(self, left, right, dmap=None):
        """numpy [B,H,W] uint8 in, the driver's 8-bit disparity images out (saturate(round_half_even(4 * D1)), stereo_vision.cpp:316)."""
        left, right, dmap, B = self._host_dmap_args(left, right, dmap)
        status = np.zeros(B, np.int32)
        self._check(lib().sv_process_batch_host_dmap(self._h, left.ctypes.data, right.ctypes.data, B, self.width, dmap.ctypes.data, status.ctypes.data))
        return dmap, status

    def submit_host_dmap(self, left, right, dmap, status=None):
        """Streaming form of process_host_dmap: returns at once; the arrays must stay alive and untouched until wait()."""
        left_c, right_c, dmap, B = self._host_dmap_args(left, right, dmap)
        if left_c is not left and not np.shares_memory(left_c, left) or right_c is not right and not np.shares_memory(right_c, right):
            raise ValueError("submit_host_dmap needs contiguous uint8 arrays (a temporary copy would be freed before the engine reads it)")
        st = status.ctypes.data_as(ctypes.c_void_p) if status is not None else None
        self._check(lib().sv_submit_batch_host_dmap(self._h, left_c.ctypes.data, right_c.ctypes.data, B, self.width, dmap.ctypes.data, st))

    def elas_process(self, I1, I2):
        """Elas::process(I1, I2, D1, D2, dims) for one pair (elas.h:153-162)."""
        I1 = np.ascontiguousarray(I1, dtype=np.uint8)
        I2 = np.ascontiguousarray(I2, dtype=np.uint8)
        H, W = I1.shape
        D1 = np.zeros((self.map_height, self.map_width), np.float32)
        D2 = np.zeros((self.map_height, self.map_width), np.float32)
        dims = (ctypes.c_int32 * 3)(W, H, W)
        self._check(lib().sv_elas_process(self._h, I1.ctypes.data, I2.ctypes.data, D1.ctypes.data, D2.ctypes.data, dims))
        return D1, D2

    def debug(self, name):
        n = lib().sv_debug_size(self._h, name.encode())
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_STAGE_DTYPES.get(name, np.float32))
        out = np.empty(n // dt.itemsize, dtype=dt)
        got = lib().sv_debug_get(self._h, name.encode(), out.ctypes.data, n)
        assert got == n
        return out

    def query(self):
        """What the handle decided at creation: host threads, chunk, slots, where the lattice filters and the triangulations run."""
        L = lib()
        keys = ["host_threads", "chunk", "slots", "gpu_lattice_filter", "gpu_triangulation"]
        out = {k: int(L.sv_query(self._h, i)) for i, k in enumerate(keys)}
        out["numa_bound"] = int(L.sv_query(self._h, 7))
        out["resident"] = int(L.sv_query(self._h, 8))
        out["host_copies"] = int(L.sv_query(self._h, 9))  # 0 not decided yet (no host-memory batch so far), 1 hipMemcpyAsync, 2 DMA lanes
        out["latency_split"] = int(L.sv_query(self._h, 10))  # single pairs: triangulations on one thread each (0), in halves (1), in quarters (2)
        return out

    def gpu_triangulation_share(self):
        """Fraction of the pairs so far whose triangulations the GPU kernel built (host mode: the dispatcher's load balancing)."""
        return int(lib().sv_query(self._h, 5)) / 1000.0

    def gpu_triangulation_fallbacks(self):
        """Vertex sets of that share which the host triangulated after all (too large for the kernels)."""
        return int(lib().sv_query(self._h, 6))

    def timing(self, on=True, only=None):
        """HIP-event timing of the kernel launches; `only` = iterable of kernel names restricts it (cheaper)."""
        if lib().sv_kernel_timing_select(self._h, ",".join(only).encode() if only else None) != 0:
            raise ValueError("unknown kernel name in %r" % (only,))
        lib().sv_kernel_timing_enable(self._h, int(on))
        lib().sv_kernel_times_reset(self._h)

    def counters(self, enable=None):
        """Work counters of the matching kernels.  counters(True) enables and resets them, counters(False) disables;
        counters() returns {dense_candidates, dense_pixels, support_energies} since the last reset."""
        L = lib()
        L.sv_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        out = (ctypes.c_uint64 * 8)()
        self._check(L.sv_debug_counters(self._h, -1 if enable is None else int(bool(enable)), out))
        return {"dense_candidates": int(out[0]), "dense_pixels": int(out[1]), "support_energies": int(out[2]),
                "dense_band_full": int(out[3]), "dense_band_partial": int(out[4]), "dense_band_per_lane": int(out[5]),
                "dense_grid_wave_trips": int(out[6]), "dense_grid_lane_trips": int(out[7])}

    def kernel_times(self):
        """{kernel: (total_ms, calls)} accumulated since timing(True)."""
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_double * cap)()
        calls = (ctypes.c_int64 * cap)()
        n = lib().sv_kernel_times(self._h, names, ms, calls, cap)
        return {names[i].decode(): (ms[i], calls[i]) for i in range(n)}


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = lib().sv_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise StereoError("sv_host_alloc(%d) failed" % nbytes)

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                lib().sv_host_free(self.ptr)
            except TypeError:  # interpreter shutdown: the module globals are already gone (the process's memory goes with it)
                pass
            self.ptr = None


def pinned_array(shape, dtype):
    """numpy array in page-locked host memory (sv_host_alloc): the engine's host path moves it by DMA without a staging copy.
    The memory lives as long as the array (or any view of it)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    block = _PinnedBlock(n)
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(block.ptr)
    buf._sv_block = block  # keeps the allocation alive: the array's base chain holds the ctypes buffer
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def reproject(disp, Q, XR=None, XT=None, want_dmap=True):
    """Batched disparity -> (u8 x4 map, 3-D points) on the device (stereo_vision.cpp:316, :233-256; optional robot-frame transform of
    the CUDA variant).  disp: CUDA float32 tensor [B,H,W]; Q: 4x4; returns (dmap uint8 [B,H,W] or None, points float64 [B,H,W,3])."""
    import torch
    assert disp.is_cuda and disp.dtype == torch.float32 and disp.dim() == 3
    disp = disp.contiguous()
    B, H, W = disp.shape
    q = np.ascontiguousarray(Q, dtype=np.float64).reshape(16)
    xr = None if XR is None else np.ascontiguousarray(XR, dtype=np.float64).reshape(9)
    xt = None if XT is None else np.ascontiguousarray(XT, dtype=np.float64).reshape(3)
    dmap = torch.empty((B, H, W), dtype=torch.uint8, device=disp.device) if want_dmap else None
    pts = torch.empty((B, H, W, 3), dtype=torch.float64, device=disp.device)
    torch.cuda.current_stream(disp.device).synchronize()
    L = lib()
    L.sv_reproject_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p]
    with torch.cuda.device(disp.device):
        rc = L.sv_reproject_batch_device(disp.data_ptr(), B, W, H, q.ctypes.data, xr.ctypes.data if xr is not None else None,
                                         xt.ctypes.data if xt is not None else None, dmap.data_ptr() if dmap is not None else None, pts.data_ptr())
    if rc != 0:
        raise StereoError("sv_reproject_batch_device failed (%d)" % rc)
    return dmap, pts


def disparity_to_u8(disp, out=None):
    """The driver's 8-bit disparity image (saturate(round_half_even(4 * d)), stereo_vision.cpp:316) of a CUDA float32 tensor, on torch's
    current stream (not waited for)."""
    import torch
    assert disp.is_cuda and disp.dtype == torch.float32 and disp.is_contiguous()
    if out is None:
        out = torch.empty(disp.shape, dtype=torch.uint8, device=disp.device)
    assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == disp.numel()
    L = lib()
    L.sv_disparity_to_u8_device.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    with torch.cuda.device(disp.device):
        rc = L.sv_disparity_to_u8_device(disp.data_ptr(), disp.numel(), out.data_ptr(), torch.cuda.current_stream(disp.device).cuda_stream)
    if rc != 0:
        raise StereoError("sv_disparity_to_u8_device failed (%d)" % rc)
    return out


def host_support_filter(params, dcan, width, height, threads=0, lattice=False):
    """Product host stage: lattice filters + corner points (CPU by design; see csrc/host_stage.h).  threads > 0: the lattice shared between
    that many threads (what single-pair calls do); lattice=True: also return the filtered lattice."""
    d = np.ascontiguousarray(dcan, dtype=np.int16).copy()
    cap = d.size + 6
    out = np.empty((cap, 3), np.int32)
    if threads > 0:
        n = lib().sv_host_support_filter_threads(ctypes.byref(params), d.ctypes.data, width, height, out.ctypes.data, cap, threads)
    else:
        n = lib().sv_host_support_filter(ctypes.byref(params), d.ctypes.data, width, height, out.ctypes.data, cap)
    if n < 0:
        raise StereoError("support capacity")
    return (out[:n].copy(), d) if lattice else out[:n].copy()


def gpu_delaunay(xy, reps=1):
    """Triangulation with the divide-and-conquer phase on the GPU (test hook).  Returns (triangles (nt,3) int32, kernel ms)."""
    share_hip_runtime_with_torch()
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    n = xy.shape[0]
    out = np.empty((2 * n + 8, 3), np.int32)
    ms = ctypes.c_double(0.0)
    L = lib()
    L.sv_gpu_delaunay.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    nt = L.sv_gpu_delaunay(xy.ctypes.data, n, out.ctypes.data, 2 * n + 8, int(reps), ctypes.byref(ms))
    if nt < 0:
        raise StereoError("sv_gpu_delaunay failed (%d): %s" % (nt, L.sv_last_error(None).decode()))
    return out[:nt].copy(), ms.value


def host_kd_order(xy):
    """Test hook: ids of the vertices that survive the duplicate scan, in the order the triangulation's recursion consumes them
    (host: radix sort / the reference's quicksort when points coincide, duplicate scan, k-d order)."""
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = np.empty(xy.shape[0], np.int32)
    L = lib()
    L.sv_host_kd_order.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    m = L.sv_host_kd_order(xy.ctypes.data, xy.shape[0], out.ctypes.data)
    if m < 0:
        raise StereoError("sv_host_kd_order failed (%d)" % m)
    return out[:m].copy()


def gpu_kd_order(xy, width, height, step, disp_max, disp=None):
    """Test hook: the same on the GPU (delaunay_gpu.hip: dg_prepare) for vertices on the support lattice of a width x height image;
    disp: the vertices' disparities (coincident vertices with equal disparity are interchangeable: the lowest id is kept).  None for
    a set the kernel leaves to the host (coincident points that are not interchangeable)."""
    share_hip_runtime_with_torch()
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    out = np.empty(xy.shape[0], np.int32)
    dp = None
    if disp is not None:
        disp = np.ascontiguousarray(disp, dtype=np.int32)
        assert disp.shape[0] == xy.shape[0]
        dp = disp.ctypes.data
    L = lib()
    L.sv_gpu_kd_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    m = L.sv_gpu_kd_order(xy.ctypes.data, dp, xy.shape[0], int(width), int(height), int(step), int(disp_max), out.ctypes.data)
    if m == -1:
        return None
    if m < 0:
        raise StereoError("sv_gpu_kd_order failed (%d)" % m)
    return out[:m].copy()


def host_delaunay(xy, split=False, helper_delay_us=0, depth=1):
    """Product host stage: Delaunay triangulation of integer points (n,2) -> (nt,3) int32.  split: build the halves (depth 2:
    quarters) of the top-level cuts on other threads (what the engine does in latency mode)."""
    xy = np.ascontiguousarray(xy, dtype=np.int32)
    n = xy.shape[0]
    out = np.empty((2 * n + 8, 3), np.int32)
    if split:
        L = lib()
        L.sv_host_delaunay_par.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        nt = L.sv_host_delaunay_par(xy.ctypes.data, n, out.ctypes.data, 2 * n + 8, int(depth), int(helper_delay_us))
    else:
        nt = lib().sv_host_delaunay(xy.ctypes.data, n, out.ctypes.data, 2 * n + 8)
    if nt < 0:
        raise StereoError("triangle capacity")
    return out[:nt].copy()
