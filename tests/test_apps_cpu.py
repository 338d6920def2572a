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


def test_damaged_clips_are_refused_not_crashed_on(built, tmp_path):
    """the clip reader of the harness programs (apps/video_io.hpp: .y4m and raw .bgr) on damaged files -- absurd or missing sizes, unknown colour
    spaces, junk tokens, overlong header lines, truncated frames, missing FRAME markers, empty files: `io_test --probe` must end with exit code 0
    (read to the end) or 2 (refused, with a message), never with a signal, and quickly."""
    rng = np.random.default_rng(5)
    good_hdr = b"YUV4MPEG2 W16 H12 F30:1 Ip A1:1 C420\n"
    frame = b"FRAME\n" + bytes(16 * 12 + 2 * 8 * 6)
    cases = {
        "empty.y4m": b"",
        "nohdr.y4m": b"\x00\x01\x02" * 50,
        "good.y4m": good_hdr + frame * 3,
        "truncated_frame.y4m": good_hdr + frame + frame[:40],
        "no_marker.y4m": good_hdr + bytes(300),
        "huge.y4m": b"YUV4MPEG2 W2000000000 H2000000000 C444\n" + frame,
        "wrap.y4m": b"YUV4MPEG2 W2147483647 H3 C420\n" + frame,
        "negative.y4m": b"YUV4MPEG2 W-16 H12 C420\n" + frame,
        "zero.y4m": b"YUV4MPEG2 W0 H0 C420\n",
        "no_size.y4m": b"YUV4MPEG2 F30:1 C420\n" + frame,
        "bad_cs.y4m": b"YUV4MPEG2 W16 H12 C411\n" + frame,
        "bad_depth.y4m": b"YUV4MPEG2 W16 H12 C420p99\n" + frame,
        "depth7.y4m": b"YUV4MPEG2 W16 H12 C420p7\n" + frame,
        "interlaced.y4m": b"YUV4MPEG2 W16 H12 It C420\n" + frame,
        "long_line.y4m": b"YUV4MPEG2 W16 H12 " + b"X" * 5000 + b" C420\n" + frame,
        "no_newline.y4m": b"YUV4MPEG2 W16 H12 C420",
        "big_ok.y4m": b"YUV4MPEG2 W30000 H30000 Cmono\n",
        "raw_bad_name.bgr": bytes(100),
        "raw_16x12.bgr": bytes(16 * 12 * 3 * 2 + 7),
        "raw_99999x3.bgr": bytes(10),
        "unknown.avi": bytes(10),
    }
    for k in range(40):                                   # random mutations of a good file
        blob = bytearray(good_hdr + frame * 2)
        for _ in range(int(rng.integers(1, 8))):
            op = int(rng.integers(0, 3))
            pos = int(rng.integers(0, min(len(blob), 60)))
            if op == 0: blob[pos] = int(rng.integers(0, 256))
            elif op == 1: del blob[pos:pos + int(rng.integers(1, 6))]
            else: blob[pos:pos] = bytes(rng.integers(32, 127, int(rng.integers(1, 12))).astype(np.uint8))
        cases["mut%02d.y4m" % k] = bytes(blob)
    seen = set()
    for name, blob in cases.items():
        path = tmp_path / name
        path.write_bytes(blob)
        r = subprocess.run([os.path.join(built, "io_test"), "--probe", str(path)], capture_output=True, timeout=30)
        out = r.stdout.decode("utf-8", "replace")               # (a refusal quotes the offending token: any bytes)
        assert r.returncode in (0, 2), (name, r.returncode, out, r.stderr)
        assert out.strip(), name
        seen.add(r.returncode)
        if name in ("good.y4m",):
            assert r.returncode == 0 and "read 3 frames" in out
        if name in ("huge.y4m", "wrap.y4m", "negative.y4m", "zero.y4m", "no_size.y4m", "bad_cs.y4m", "bad_depth.y4m", "depth7.y4m", "interlaced.y4m",
                    "empty.y4m", "nohdr.y4m", "truncated_frame.y4m", "no_marker.y4m", "raw_bad_name.bgr", "raw_99999x3.bgr", "unknown.avi"):
            assert r.returncode == 2, (name, out)
    assert seen == {0, 2}
