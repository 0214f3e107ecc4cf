"""Counterpart of the reference's Python entry point (reference: stereo_vision/sv.py) over libstereo_vision_hip.so.

Same class, constructor arguments, ctypes signature and CLI flags as the reference:

    class stereo_vision(so_lib_path, width, height, defaultCalibFile, objectTracking, graphics, display, scale,
                        pc_extrapolation, YOLO_CFG, YOLO_WEIGHTS, YOLO_CLASSES, CAMERA_CALIBRATION_YAML, subsampling)
        .generatePointCloud(left_bgr, right_bgr) -> ndarray (width*height, 3) float64      (sv.py:156-189)
    main()  argparse CLI                                                                    (sv.py:195-331)

Differences, all forced by the environment or by bugs of the reference:
  * no cv2: BGR->BGRA is a numpy concatenate, images are read with PIL;
  * the C function is called with the reference's 14 arguments (sv.py:180,189); the library does not read arguments 15/16
    (removeSky, subsampling) - half-resolution mode is selected with sv_legacy_set_subsampling() before the first frame;
  * __del__ calls clean(), which here frees the library state but does NOT exit() the interpreter;
  * the dataset download helpers of the reference's CLI are not provided (no network); --demo reads --kitti.
"""
import argparse
import ctypes
import glob
import os

import numpy as np
from numpy.ctypeslib import ndpointer

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_STEREO_VISION_SO_PATH = os.path.join(os.path.dirname(HERE), "libstereo_vision_hip.so")
DEFAULT_CALIBRATION = os.path.join(HERE, "data", "kitti_2011_09_26.yml")


class stereo_vision:
    def __init__(self, so_lib_path=DEFAULT_STEREO_VISION_SO_PATH, width=1242, height=375, defaultCalibFile=True, objectTracking=True,
                 graphics=False, display=False, scale=1, pc_extrapolation=1, YOLO_CFG="src/yolo/yolov4-tiny.cfg",
                 YOLO_WEIGHTS="src/yolo/yolov4-tiny.weights", YOLO_CLASSES="src/yolo/classes.txt",
                 CAMERA_CALIBRATION_YAML=DEFAULT_CALIBRATION, subsampling=False):
        if not os.path.exists(so_lib_path):
            raise FileNotFoundError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % so_lib_path)
        try:  # share one HIP runtime with PyTorch if the application also uses it (see engine.share_hip_runtime_with_torch)
            import torch  # noqa: F401
        except ImportError:
            pass
        self.sv = ctypes.CDLL(so_lib_path)
        self.width = width
        self.height = height
        self.sv.generatePointCloud.restype = ndpointer(dtype=ctypes.c_double, shape=(width * height, 3))
        self.defaultCalibFile = defaultCalibFile
        self.objectTracking = objectTracking
        self.graphics = graphics
        self.display = display
        self.scale = scale
        self.pc_extrapolation = pc_extrapolation
        self.YOLO_CFG = YOLO_CFG
        self.YOLO_WEIGHTS = YOLO_WEIGHTS
        self.YOLO_CLASSES = YOLO_CLASSES
        self.CAMERA_CALIBRATION_YAML = CAMERA_CALIBRATION_YAML
        self.subsampling = bool(subsampling)
        # reference: sv.py:180 - 14 entries (the C function declares 16, stereo_vision.cpp:566-581; the last two are never read here)
        self.sv.generatePointCloud.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_bool,
                                               ctypes.c_bool, ctypes.c_bool, ctypes.c_bool, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                               ctypes.c_char_p, ctypes.c_char_p]
        self.sv.sv_legacy_set_subsampling.argtypes = [ctypes.c_int]
        self.sv.sv_legacy_set_subsampling.restype = None
        self.sv.sv_legacy_set_subsampling(int(self.subsampling))  # frozen by the first frame, like the driver's static Elas (stereo_vision.cpp:307-311)
        self.sv.clean.restype = None
        self._closed = False
        self._bgra = None

    def generatePointCloud(self, left, right):
        left, right = np.asarray(left), np.asarray(right)
        if left.ndim != 3 or left.shape[2] != 3 or left.shape[:2] != (self.height, self.width) or right.shape != left.shape:
            raise ValueError("expected two BGR uint8 images of shape (%d, %d, 3)" % (self.height, self.width))
        # cv2.COLOR_BGR2BGRA: a fourth channel of 255.  Pillow's C loop does it in ~0.3 ms per image (the channel order is left
        # alone: "RGB" -> "RGBA" only appends alpha); plain numpy needs ~1.5 ms for the strided copy.  The library reads the
        # buffers during the call only.
        try:
            from PIL import Image
            self._bgra = (np.asarray(Image.fromarray(np.ascontiguousarray(left, dtype=np.uint8), "RGB").convert("RGBA")),
                          np.asarray(Image.fromarray(np.ascontiguousarray(right, dtype=np.uint8), "RGB").convert("RGBA")))
            self._bgra = tuple(np.ascontiguousarray(b) for b in self._bgra)
        except ImportError:
            if self._bgra is None or not self._bgra[0].flags.writeable:
                self._bgra = (np.full(left.shape[:2] + (4,), 255, np.uint8), np.full(left.shape[:2] + (4,), 255, np.uint8))
            self._bgra[0][:, :, :3] = left
            self._bgra[1][:, :, :3] = right
        try:
            return self.sv.generatePointCloud(self._bgra[0].ctypes.data, self._bgra[1].ctypes.data, self.CAMERA_CALIBRATION_YAML.encode("utf-8"), self.width, self.height, self.defaultCalibFile,
                                              self.objectTracking, self.graphics, self.display, self.scale, self.pc_extrapolation,
                                              self.YOLO_CFG.encode("utf-8"), self.YOLO_WEIGHTS.encode("utf-8"), self.YOLO_CLASSES.encode("utf-8"))
        except ValueError as e:  # NULL pointer from the library: initialisation or a HIP call failed (message on stderr)
            raise RuntimeError("generatePointCloud failed (see stderr)") from e

    def last_disparity_u8(self):
        """The reference's `dmap` of the last frame: uint8 (height, width), 4 x disparity (stereo_vision.cpp:316)."""
        w, h = ctypes.c_int(), ctypes.c_int()
        self.sv.sv_legacy_last_dmap.restype = ctypes.POINTER(ctypes.c_ubyte)
        p = self.sv.sv_legacy_last_dmap(ctypes.byref(w), ctypes.byref(h))
        return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()

    def object_positions(self, boxes):
        """Mean (X, Y, Z) of the last frame's cloud inside each detector box [(x, y, w, h), ...] - what the reference's
        publishPointCloud computes for its tracked objects (stereo_vision.cpp:261-278); boxes come from your own detector."""
        b = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        out = np.zeros((b.shape[0], 3), np.float64)
        self.sv.sv_legacy_box_means.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        if self.sv.sv_legacy_box_means(b.ctypes.data, b.shape[0], out.ctypes.data) != 0:
            raise RuntimeError("no frame has been processed yet")
        return out

    def close(self):
        if not self._closed:
            self._closed = True
            self.sv.clean()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _imread_bgr(path, scale):
    from PIL import Image
    im = Image.open(path).convert("RGB")
    if scale != 1:
        im = im.resize((im.width // scale, im.height // scale), Image.BILINEAR)
    return np.asarray(im)[:, :, ::-1].copy()


def main(argv=None):
    parser = argparse.ArgumentParser(description="stereo_vision CLI for disparity calculation and 3D depth map generation from a stereo pair")
    parser.add_argument("-k", "--kitti", type=str, default="~/KITTI", help="Path to KITTI directory of test images")
    parser.add_argument("-s", "--subsampling", type=int, default=0, help="Set s=1 for evaluating only every second pixel")
    parser.add_argument("-f", "--scale", type=int, default=1, help="By what factor to scale down the image by")
    parser.add_argument("-p", "--pointcloud_interpolation", default=False, action="store_true", help="Interpolates the point cloud to the desired scale")
    parser.add_argument("-prl", "--parallel", default=False, action="store_true", help="Run parallel (this library is always the GPU build)")
    parser.add_argument("-d", "--demo", default=False, action="store_true", help="Run over the image_02/image_03 folders under --kitti")
    parser.add_argument("-dst", "--dataset", choices=["kitti2015", "kitti_smol"], default="kitti_smol", help="Dataset layout under --kitti")
    parser.add_argument("-c", "--camera_calibration", type=str, default=DEFAULT_CALIBRATION, help="OpenCV YAML calibration file")
    parser.add_argument("-o", "--object_track", default=False, action="store_true", help="(accepted for compatibility; no detector in this library)")
    parser.add_argument("-ycfg", "--yolo_cfg", type=str, default="", help="YOLO CFG file")
    parser.add_argument("-yw", "--yolo_weights", type=str, default="", help="YOLO Weights file")
    parser.add_argument("-ycl", "--yolo_classes", type=str, default="", help="YOLO Classes to track")
    parser.add_argument("-ctu", "--camera_to_use", default=-1, type=int, help="(cameras need cv2; not available)")
    parser.add_argument("-sw", "--swap", default=False, action="store_true", help="Swaps cameras")
    parser.add_argument("-n", "--frames", type=int, default=0, help="stop after this many frames (0 = all)")
    args = parser.parse_args(argv)

    root = os.path.expanduser(args.kitti)
    if args.dataset == "kitti2015":
        ldir, rdir = os.path.join(root, "testing", "image_2"), os.path.join(root, "testing", "image_3")
    else:
        ldir, rdir = os.path.join(root, "image_02"), os.path.join(root, "image_03")
        if os.path.isdir(os.path.join(ldir, "data")):
            ldir, rdir = os.path.join(ldir, "data"), os.path.join(rdir, "data")
    files = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ldir, "*.png")))
    if not files:
        parser.error("no PNG images under %s" % ldir)
    s = stereo_vision(width=1242 // args.scale, height=375 // args.scale, objectTracking=args.object_track, display=False, graphics=False,
                      scale=args.scale, pc_extrapolation=int(args.pointcloud_interpolation), CAMERA_CALIBRATION_YAML=args.camera_calibration,
                      subsampling=bool(args.subsampling))
    import time
    n = 0
    for name in files:
        left, right = _imread_bgr(os.path.join(ldir, name), args.scale), _imread_bgr(os.path.join(rdir, name), args.scale)
        if args.swap:
            left, right = right, left
        t0 = time.perf_counter()
        pts = s.generatePointCloud(left, right)
        dt = time.perf_counter() - t0
        print("(FPS=%f) (%d, %d) (t_t=%f) valid=%.3f" % (1.0 / dt, s.height, s.width, dt, float(np.isfinite(pts[:, 2]).mean())))
        n += 1
        if args.frames and n >= args.frames:
            break
    s.close()
