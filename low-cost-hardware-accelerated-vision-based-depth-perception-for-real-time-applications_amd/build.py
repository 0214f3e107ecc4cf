"""In-tree build of libstereo_vision_hip.so (hipcc, gfx950 only).

    python build.py            # incremental
    python build.py --force

Flags that matter for parity: -ffp-contract=off (the reference's serial path has no FMA) and
correctly rounded fp32 divide (plane/edge-line arithmetic of elas.cpp:894-906 must match x86 IEEE results).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstereo_vision_hip.so")
SOURCES = ["kernels.hip", "delaunay_gpu.hip", "legacy_kernels.hip", "engine.cpp", "host_stage.cpp", "legacy.cpp", "calib.cpp", "dma_lanes.cpp"]
HEADERS = ["sv_kernels.h", "host_stage.h", "calib.h", "dma_lanes.h", os.path.join("..", "..", "include", "stereo_vision_hip.h")]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
HOST = ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
DEVICE = ["--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + COMMON + (DEVICE if src.endswith(".hip") else HOST) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
