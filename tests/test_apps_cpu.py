"""Harness programs (apps/): CPU-side checks -- clip I/O and the jitter statistic through the
C++ unit test, and that the programs refuse to run without a GPU instead of falling back to anything."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APPS = os.path.join(ROOT, "apps")
BIN = os.path.join(APPS, "bin")
PROGRAMS = ["vs_video_test", "vs_eval_jitter", "vs_grid_search_align", "vs_grid_search_smoother"]


@pytest.fixture(scope="module")
def built(vs):
    subprocess.check_call(["make", "-C", APPS, "-s", "-j4"])
    return BIN


def test_io_and_jitter_unit_test(built, tmp_path):
    r = subprocess.run([os.path.join(built, "io_test"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok ")


def test_programs_are_built(built):
    for p in PROGRAMS:
        assert os.access(os.path.join(built, p), os.X_OK), p


@pytest.mark.parametrize("prog", ["vs_eval_jitter", "vs_grid_search_align", "vs_grid_search_smoother"])
def test_usage_without_arguments(built, prog):
    # eval_jitter.cpp:23-26, grid_search_align.cpp:64-67: usage on stderr, exit code 1
    r = subprocess.run([os.path.join(built, prog)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Usage:" in r.stderr


def test_video_test_missing_input_dir(built, tmp_path):
    r = subprocess.run([os.path.join(built, "vs_video_test"), str(tmp_path / "nope"), str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Input directory does not exist" in r.stderr
    # the output directory is created first, like the reference (video_test.cpp:16-26)
    assert (tmp_path / "out").is_dir()


def test_video_test_no_clips(built, tmp_path):
    (tmp_path / "in").mkdir()
    (tmp_path / "in" / "notes.txt").write_text("x")
    r = subprocess.run([os.path.join(built, "vs_video_test"), str(tmp_path / "in"), str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "No .y4m / .bgr clips found" in r.stderr


def test_no_gpu_is_a_loud_failure(built, tmp_path, vs):
    if vs.device_count() > 0:
        pytest.skip("a GPU is present; covered by the gpu tests")
    (tmp_path / "in").mkdir()
    np.zeros((2, 16, 16, 3), np.uint8).tofile(tmp_path / "in" / "clip_16x16.bgr")
    r = subprocess.run([os.path.join(built, "vs_video_test"), str(tmp_path / "in"), str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "no HIP device" in r.stderr
    r = subprocess.run([os.path.join(built, "vs_grid_search_align"), str(tmp_path / "in" / "clip_16x16.bgr")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "No HIP device" in r.stderr
