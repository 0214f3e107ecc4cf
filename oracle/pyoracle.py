"""TEST INFRASTRUCTURE — ctypes front-end for the two CPU checkers.

* ``Oracle``  : our CPU restatement (oracle/liboracle.so, built from oracle/elas_oracle.cpp)
* ``RefElas`` : the reference's own serial LIBELAS (oracle/_ref/libelas_ref.so, built in the
                container from /root/reference by oracle/Makefile; may be absent elsewhere)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class ElasParams(ctypes.Structure):
    """Mirror of oracle/elas_params.h (== Elas::parameters, reference elas.h:60-145)."""

    _fields_ = [
        ("disp_min", ctypes.c_int32),
        ("disp_max", ctypes.c_int32),
        ("support_threshold", ctypes.c_float),
        ("support_texture", ctypes.c_int32),
        ("candidate_stepsize", ctypes.c_int32),
        ("incon_window_size", ctypes.c_int32),
        ("incon_threshold", ctypes.c_int32),
        ("incon_min_support", ctypes.c_int32),
        ("add_corners", ctypes.c_int32),
        ("grid_size", ctypes.c_int32),
        ("beta", ctypes.c_float),
        ("gamma", ctypes.c_float),
        ("sigma", ctypes.c_float),
        ("sradius", ctypes.c_float),
        ("match_texture", ctypes.c_int32),
        ("lr_threshold", ctypes.c_int32),
        ("speckle_sim_threshold", ctypes.c_float),
        ("speckle_size", ctypes.c_int32),
        ("ipol_gap_width", ctypes.c_int32),
        ("filter_median", ctypes.c_int32),
        ("filter_adaptive_mean", ctypes.c_int32),
        ("postprocess_only_left", ctypes.c_int32),
        ("subsampling", ctypes.c_int32),
    ]

    @classmethod
    def preset(cls, setting):
        """setting: 'robotics' | 'middlebury' (elas.h:92-143)."""
        p = cls()
        p.disp_min, p.disp_max = 0, 255
        p.support_texture, p.candidate_stepsize = 10, 5
        p.incon_window_size, p.incon_threshold, p.incon_min_support = 5, 5, 5
        p.grid_size, p.beta, p.sigma = 20, 0.02, 1.0
        p.lr_threshold, p.speckle_sim_threshold, p.speckle_size = 2, 1.0, 200
        p.subsampling = 0
        if setting == "robotics":
            p.support_threshold, p.add_corners, p.gamma, p.sradius = 0.85, 0, 3.0, 2.0
            p.match_texture, p.ipol_gap_width = 1, 3
            p.filter_median, p.filter_adaptive_mean, p.postprocess_only_left = 0, 1, 1
        elif setting == "middlebury":
            p.support_threshold, p.add_corners, p.gamma, p.sradius = 0.95, 1, 5.0, 3.0
            p.match_texture, p.ipol_gap_width = 0, 5000
            p.filter_median, p.filter_adaptive_mean, p.postprocess_only_left = 1, 0, 0
        else:
            raise ValueError(setting)
        return p

    @classmethod
    def driver(cls, disp_max=255):
        """What the reference driver runs (stereo_vision.cpp:307-311)."""
        p = cls.preset("middlebury")
        p.postprocess_only_left = 1
        p.filter_adaptive_mean = 1
        p.disp_max = disp_max
        return p


_STAGE_DTYPES = {
    "desc1": np.uint8, "desc2": np.uint8,
    "dcan_raw": np.int16, "dcan_dims": np.int32,
    "support": np.int32, "tri1": np.int32, "tri2": np.int32,
    "planes1": np.float32, "planes2": np.float32,
    "grid1": np.int32, "grid2": np.int32, "grid_dims": np.int32,
}


def _stage_dtype(name):
    return _STAGE_DTYPES.get(name, np.float32)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


class _StageLib:
    """Common `*_run_stages / *_size / *_get` protocol."""

    prefix = None

    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)
        L, pf = self.lib, self.prefix
        self._run = getattr(L, pf + "_run_stages")
        self._run.restype = ctypes.c_int
        self._run.argtypes = [ctypes.POINTER(ElasParams), ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8),
                              ctypes.c_int, ctypes.c_int, ctypes.c_int]
        self._size = getattr(L, pf + "_size")
        self._size.restype = ctypes.c_long
        self._size.argtypes = [ctypes.c_char_p]
        self._get = getattr(L, pf + "_get")
        self._get.restype = ctypes.c_long
        self._get.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long]
        self._proc = getattr(L, pf + "_process")
        self._proc.restype = ctypes.c_double
        self._proc.argtypes = [ctypes.POINTER(ElasParams), ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8),
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                               ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.c_int]
        self._delaunay = getattr(L, pf + "_delaunay")
        self._delaunay.restype = ctypes.c_int
        self._delaunay.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]

    def run_stages(self, params, left, right):
        """Run the whole pipeline keeping every intermediate; returns #support points."""
        H, W = left.shape
        l, lp = _u8(left)
        r, rp = _u8(right)
        n = self._run(ctypes.byref(params), lp, rp, W, H, W)
        if n < 0:
            raise RuntimeError("%s_run_stages failed: %d" % (self.prefix, n))
        return n

    def stage(self, name, shape=None):
        n = self._size(name.encode())
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_stage_dtype(name))
        out = np.empty(n // dt.itemsize, dtype=dt)
        got = self._get(name.encode(), out.ctypes.data_as(ctypes.c_void_p), n)
        assert got == n
        return out.reshape(shape) if shape is not None else out

    def stages(self, names):
        return {k: self.stage(k) for k in names}

    def process(self, params, left, right, canonical=True, reps=1):
        """Operator-seam call (Elas::process semantics).  Returns (D1, D2, seconds_per_call)."""
        H, W = left.shape
        l, lp = _u8(left)
        r, rp = _u8(right)
        Hm, Wm = (H // 2, W // 2) if params.subsampling else (H, W)  # elas.h:160-161: half-size maps when subsampling
        D1 = np.zeros((Hm, Wm), np.float32)
        D2 = np.zeros((Hm, Wm), np.float32)
        t = self._proc(ctypes.byref(params), lp, rp, W, H, W, D1.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                       D2.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(bool(canonical)), int(reps))
        return D1, D2, t

    def delaunay(self, xy):
        """xy: (n,2) float32 points -> (nt,3) int32 triangles in the reference's output order."""
        xy = np.ascontiguousarray(xy, dtype=np.float32)
        n = xy.shape[0]
        cap = 2 * n + 16
        out = np.empty((cap, 3), np.int32)
        nt = self._delaunay(xy.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), cap)
        assert 0 <= nt <= cap
        return out[:nt].copy()


class RefElas(_StageLib):
    prefix = "ref"
    default_path = os.path.join(HERE, "_ref", "libelas_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.default_path)

    def __init__(self, path=None):
        super().__init__(path or self.default_path)


class Oracle(_StageLib):
    prefix = "orc"
    default_path = os.path.join(HERE, "liboracle.so")

    def __init__(self, path=None):
        path = path or self.default_path
        if not os.path.exists(path):
            build()
        super().__init__(path)


def build(ref=True):
    """Compile the checker(s).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
