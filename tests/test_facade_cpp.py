"""The C++ drop-in facade (video_stabilizer_amd/facade/*.hpp: the reference's imgproc.hpp / alignment.hpp /
stabilizer.hpp / smoother.hpp names on the C ABI), exercised by a C++ rewrite of the reference's align_test.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = "/tmp/vs_facade_test_%d" % os.getpid()


EXE_MAT = "/tmp/vs_facade_cvmat_test_%d" % os.getpid()


def _build(src="facade_test.cpp", exe=EXE, extra=()):
    lib = os.path.join(ROOT, "video_stabilizer_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", src), *extra, "-L" + lib, "-lvs_amd",
           "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def _build_cvmat():
    # the cv::Mat overloads (alignment.hpp:55-58, stabilizer.hpp:39, imgproc.hpp:97) against the cv::Mat test double
    _build("facade_cvmat_test.cpp", EXE_MAT, ["-I" + os.path.join(ROOT, "tests", "cpp", "stubs")])


def test_facade_transform_algebra_cpu(vs):
    _build()
    out = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_facade_align_pair_and_stabilizer_gpu(gpu_vs):
    _build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr


def test_facade_cvmat_overloads_compile_and_check_types_cpu(vs):
    _build_cvmat()
    out = subprocess.run([EXE_MAT, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_facade_cvmat_overloads_equal_the_pointer_forms_gpu(gpu_vs):
    _build_cvmat()
    out = subprocess.run([EXE_MAT, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL PASS" in out.stdout, out.stdout + out.stderr
