"""Host stage under AddressSanitizer + UBSan and under ThreadSanitizer (CPU build; the GPU pool offers no sanitizers)."""
import os
import shutil
import subprocess

import pytest

import util

CSRC = os.path.join(util.ROOT, util.PKG, "csrc")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_stage_sanitizers(tmp_path, san):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "san_host")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-mavx2", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-ffp-contract=off",
           "-I" + os.path.join(util.ROOT, "include"), "-I" + CSRC, os.path.join(util.HERE, "san_host.cpp"), os.path.join(CSRC, "host_stage.cpp"),
           "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in (r.stderr or "").lower() and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_gpu_delaunay_logic_emulated_on_cpu(tmp_path):
    """The device functions of csrc/delaunay_gpu.hip (leaf construction, merge, tree / slot arithmetic) compiled as plain C++ and
    run depth by depth with the nodes of a depth in reversed order, against Delaunay::triangulate, under ASan + UBSan."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "emu_dg")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-mavx2", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-x", "c++",
           "-I" + os.path.join(util.ROOT, "include"), "-I" + CSRC, os.path.join(util.HERE, "emu_delaunay_gpu.cpp"), os.path.join(CSRC, "host_stage.cpp"),
           "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in (r.stderr or ""):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_gpu_vertex_preparation_emulated_on_cpu(tmp_path):
    """The workgroup-cooperative preparation of csrc/delaunay_gpu.hip (dg_prepare in LDS, dg_prepare_global in scratch: bit maps, u16
    rank prefixes, in-place partitions) compiled unchanged and run as one fiber per GPU thread under ASan + UBSan, in buffers of exactly
    the size the launchers request, against Delaunay::kd_ordered_ids: the corner row, negative x on the right side, coincident groups up
    to the limit and beyond, DG_PREP_MAX vertices and one more, a vertex off the bit maps, 4K-sized sets with 1 024 threads; every
    thread of a workgroup must return the same value after the same number of barriers."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "emu_prep")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-mavx2", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-x", "c++",
           "-I" + os.path.join(util.ROOT, "include"), "-I" + CSRC, os.path.join(util.HERE, "emu_dg_prepare.cpp"), os.path.join(CSRC, "host_stage.cpp"),
           "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in (r.stderr or ""):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-500:], r.stderr[-3000:])
