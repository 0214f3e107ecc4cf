"""Shared helpers for the test-suite (fixtures on disk, case parameters, package import)."""
import hashlib
import importlib
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
PKG = "low-cost-hardware-accelerated-vision-based-depth-perception-for-real-time-applications_amd"

STAGES = ["desc1", "desc2", "dcan_raw", "support", "tri1", "tri2", "planes1", "planes2", "grid1", "grid2",
          "wta1", "wta2", "lr1", "lr2", "speckle1", "speckle2", "gap1", "gap2", "amean1", "amean2", "final1", "final2"]


def pkg(sub=None):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        return json.load(f)


def load_png(name):
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(GOLDEN, name))))


def case_images(entry):
    """Input pair of a digests.json entry."""
    if "synth" in entry:
        return pkg("synth").make_pair(**entry["synth"])
    img = entry["image"]
    if img.startswith("kitti") and img[5:].isdigit():
        return load_png(img + "_left.png"), load_png(img + "_right.png")
    if img == "kitti0_crop":
        l, r = load_png("kitti0_left.png"), load_png("kitti0_right.png")
        return l[150:278, 400:720].copy(), r[150:278, 400:720].copy()
    if img == "cones_crop":
        return load_png("cones_crop_left.png"), load_png("cones_crop_right.png")
    raise KeyError(img)


def case_params(entry, cls):
    """cls: a ctypes params Structure class with .driver/.preset constructors."""
    if entry["preset"] == "driver":
        p = cls.driver(entry["disp_max"])
    else:
        p = cls.preset(entry["preset"])
        p.disp_max = entry["disp_max"]
    p.subsampling = 1 if entry.get("subsampling") else 0
    p.disp_min = int(entry.get("disp_min", 0))
    return p


def map_shape(entry):
    """(rows, cols) of the disparity maps of a case: half the image size in half-resolution mode (elas.h:160-161)."""
    h, w = entry["shape"]
    return (h // 2, w // 2) if entry.get("subsampling") else (h, w)


def golden_npz(name):
    path = os.path.join(GOLDEN, name + ".npz")
    return dict(np.load(path)) if os.path.exists(path) else {}
