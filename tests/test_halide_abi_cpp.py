"""The link-level drop-in (include/vs_halide_abi.h, libvs_halide_abi.so): the reference's sixteen Halide AOT symbols with
halide_buffer_t arguments, driven by a C++ program shaped like the reference's imgproc.cpp wrappers (tests/cpp/halide_abi_test.cpp)
and compared with the oracle byte for byte."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/vs_halide_abi_test_%d" % os.getpid()
LIB = os.path.join(ROOT, "video_stabilizer_amd")
AOT_SYMBOLS = ["pyr_down", "grad_xy", "sparse_jac", "sparse_warpdiff", "sparse_ica", "image_warp"] + ["grad_argmax_%d" % t for t in range(2, 21, 2)]


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "halide_abi_test.cpp"),
           "-L" + LIB, "-lvs_halide_abi", "-lvs_amd", "-Wl,-rpath," + LIB, "-L" + os.path.join(ROOT, "oracle"), "-lvs_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_shim_exports_the_sixteen_aot_symbols(vs):
    lib = ctypes.CDLL(os.path.join(LIB, "libvs_halide_abi.so"))
    for name in AOT_SYMBOLS:
        assert hasattr(lib, name), name
    assert len(AOT_SYMBOLS) == 16          # imgproc.cpp:9-24 includes sixteen generated headers
    # every function include/vs_halide_abi.h declares is one of them, and all of them are declared
    import re
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vs_halide_abi.h")).read(), flags=re.S)
    assert set(re.findall(r"^int (\w+)\(", text, flags=re.M)) == set(AOT_SYMBOLS)
    assert "oracle" not in os.popen("ldd %s 2>/dev/null" % os.path.join(LIB, "libvs_halide_abi.so")).read()
    # ... and the product library itself does NOT export those generic names (they live in the shim only)
    core = ctypes.CDLL(os.path.join(LIB, "libvs_amd.so"))
    assert not any(hasattr(core, n) for n in AOT_SYMBOLS)


def test_shim_argument_validation_cpu(vs):
    _build()
    out = subprocess.run([EXE, "args"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_reference_shaped_wrappers_on_the_shim_equal_the_oracle_gpu(gpu_vs):
    _build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL PASS (gpu)" in out.stdout, out.stdout + out.stderr
