"""Harness programs on the GPU (apps/): the file driver must write exactly what the library's
stabilizer returns for the decoded frames, the jitter tool must print the statistic of the library's own transforms,
and the grid searches must run every combination on a device-resident clip."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "apps", "bin")
W, H, N = 320, 240, 48


def run(prog, *args, timeout=600):
    r = subprocess.run([os.path.join(BIN, prog), *map(str, args)], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (prog, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def flow_median(t, w, h, lattice=33):
    """python twin of apps/jitter.hpp flow_median"""
    i = (np.arange(lattice) + 0.5)
    u = (i * w / lattice - 0.5 * w)[None, :]
    v = (i * h / lattice - 0.5 * h)[:, None]
    dx = t.A * u - t.B * v + t.TX
    dy = t.B * u + t.A * v + t.TY
    mag = np.sort(np.sqrt(dx * dx + dy * dy).astype(np.float32).ravel())
    return float(mag[mag.size // 2])


def jitter_of(vs, frames):
    _, ts = vs.Aligner(device=0).align_batch(frames)
    return float(np.median([flow_median(t, frames.shape[2], frames.shape[1]) for t in ts[1:]]))


def bgr_to_y4m(path, frames, chroma):
    """numpy twin of apps/video_io.hpp Writer (8 bit)"""
    n, h, w, _ = frames.shape
    b, g, r = (frames[..., c].astype(np.int32) for c in range(3))
    y = np.clip(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16, 0, 255)
    cb = np.clip(((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128, 0, 255)
    cr = np.clip(((112 * r - 94 * g - 18 * b + 128) >> 8) + 128, 0, 255)
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C%s\n" % (w, h, b"444" if chroma == 444 else b"420jpeg"))
        for k in range(n):
            f.write(b"FRAME\n")
            f.write(y[k].astype(np.uint8).tobytes())
            for p in (cb[k], cr[k]):
                if chroma == 420:
                    p = (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) // 4
                f.write(p.astype(np.uint8).tobytes())


def y4m_to_bgr(path):
    """numpy twin of apps/video_io.hpp Reader (8 bit 4:2:0 / 4:4:4) -> (frames, header)"""
    raw = open(path, "rb").read()
    nl = raw.index(b"\n")
    header = raw[:nl].decode()
    w = int(re.search(r" W(\d+)", header).group(1))
    h = int(re.search(r" H(\d+)", header).group(1))
    c444 = " C444" in header
    cw, ch = (w, h) if c444 else ((w + 1) // 2, (h + 1) // 2)
    fsz = w * h + 2 * cw * ch
    pos, frames = nl + 1, []
    while pos < len(raw):
        assert raw[pos:pos + 6] == b"FRAME\n"
        pos += 6
        a = np.frombuffer(raw, np.uint8, fsz, pos).astype(np.int32)
        pos += fsz
        y = a[:w * h].reshape(h, w)
        u = a[w * h:w * h + cw * ch].reshape(ch, cw)
        v = a[w * h + cw * ch:].reshape(ch, cw)
        if not c444:
            u = np.repeat(np.repeat(u, 2, 0), 2, 1)[:h, :w]
            v = np.repeat(np.repeat(v, 2, 0), 2, 1)[:h, :w]
        c, d, e = y - 16, u - 128, v - 128
        r = np.clip((298 * c + 409 * e + 128) >> 8, 0, 255)
        g = np.clip((298 * c - 100 * d - 208 * e + 128) >> 8, 0, 255)
        b = np.clip((298 * c + 516 * d + 128) >> 8, 0, 255)
        frames.append(np.stack([b, g, r], -1).astype(np.uint8))
    return np.stack(frames), header


@pytest.fixture(scope="module")
def clip(gpu_vs, tmp_path_factory):
    from video_stabilizer_amd import synth
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "apps"), "-s", "-j4"])
    d = tmp_path_factory.mktemp("recordings")
    frames, _ = synth.make_clip(W, H, N, seed=77, channels=3)
    raw = d / ("shaky_%dx%d.bgr" % (W, H))
    frames.tofile(raw)
    bgr_to_y4m(d / "shaky420.y4m", frames, 420)
    return d, frames, raw


def test_eval_jitter_prints_the_library_statistic(gpu_vs, clip):
    d, frames, raw = clip
    out = run("vs_eval_jitter", raw, d / "shaky420.y4m", d / "missing.y4m")
    lines = dict(l.split("\tmedian_jitter_px=") for l in out.strip().splitlines())
    assert set(lines) == {str(raw), str(d / "shaky420.y4m")}          # the unreadable clip is reported and skipped
    want = jitter_of(gpu_vs, frames)
    assert want > 1.0                                                   # the synthetic camera shakes by several pixels
    assert abs(float(lines[str(raw)]) - want) <= 1e-5 * want + 1e-6
    decoded, _ = y4m_to_bgr(d / "shaky420.y4m")
    want420 = jitter_of(gpu_vs, decoded)
    assert abs(float(lines[str(d / "shaky420.y4m")]) - want420) <= 1e-5 * want420 + 1e-6


def test_video_test_writes_what_the_stabilizer_returns(gpu_vs, clip, tmp_path):
    d, frames, raw = clip
    out = run("vs_video_test", d, tmp_path / "out", "--chunk", 20)      # 48 frames = chunks of 20 + 20 + 8
    assert "All videos have been processed successfully." in out
    assert "Finished processing 48 frames (38 written" in out           # lag = 10 frames of latency, no flush (stabilizer.cpp:46-52)

    stab = gpu_vs.Stabilizer(device=0, crop_pixels=0)
    want = [o for o in (stab.process(f) for f in frames) if o is not None]
    got = np.fromfile(tmp_path / "out" / ("processed_" + raw.name), np.uint8).reshape(-1, H, W, 3)
    assert got.shape[0] == len(want) == N - 10
    assert np.array_equal(got, np.stack(want))

    # the .y4m clip: decoded with the twin reader, stabilized by the library, encoded with the twin writer == the file
    decoded, _ = y4m_to_bgr(d / "shaky420.y4m")
    stab = gpu_vs.Stabilizer(device=0, crop_pixels=0)
    want = np.stack([o for o in (stab.process(f) for f in decoded) if o is not None])
    bgr_to_y4m(tmp_path / "want.y4m", want, 420)
    assert open(tmp_path / "out" / "processed_shaky420.y4m", "rb").read() == open(tmp_path / "want.y4m", "rb").read()

    # and the point of it all: the output shakes less than the input
    j_in = float(run("vs_eval_jitter", raw).split("=")[1])
    j_out = float(run("vs_eval_jitter", tmp_path / "out" / ("processed_" + raw.name)).split("=")[1])
    assert j_out < 0.6 * j_in, (j_in, j_out)


def test_video_test_crop_and_bilinear(gpu_vs, clip, tmp_path):
    d, frames, raw = clip
    only = tmp_path / "in"
    only.mkdir()
    os.symlink(raw, only / raw.name)
    run("vs_video_test", only, tmp_path / "out", "--crop", 16, "--bilinear")
    stab = gpu_vs.Stabilizer(device=0, crop_pixels=16, warp_mode=gpu_vs.WARP_BILINEAR)
    want = np.stack([o for o in (stab.process(f) for f in frames) if o is not None])
    got = np.fromfile(tmp_path / "out" / ("processed_" + raw.name), np.uint8).reshape(-1, H - 32, W - 32, 3)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("flag,mode", [("--bilinear", "WARP_BILINEAR"), ("--lanczos2", "WARP_LANCZOS2")])
def test_video_test_output_matches_the_oracle_stabilizer(gpu_vs, oracle, clip, tmp_path, flag, mode):
    """Parity, not plumbing: the file the harness writes against the CPU oracle's VideoStabilizer run on the same frames
    (transforms agree to 1e-4 px, so a pixel may land on the other side of a rounding boundary: <= 1 LSB, almost all equal)."""
    d, frames, raw = clip
    only = tmp_path / "in"
    only.mkdir()
    os.symlink(raw, only / raw.name)
    run("vs_video_test", only, tmp_path / "out", "--crop", 16, "--chunk", 13, flag)
    got = np.fromfile(tmp_path / "out" / ("processed_" + raw.name), np.uint8).reshape(-1, H - 32, W - 32, 3)
    cpu = oracle.Stabilizer(crop_pixels=16, warp_mode=getattr(oracle, mode))
    want = np.stack([o for o in (cpu.process(f) for f in frames) if o is not None])
    assert got.shape == want.shape
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-2, (int(diff.max()), float((diff != 0).mean()))


def test_grid_search_align(gpu_vs, clip):
    d, frames, raw = clip
    out = run("vs_grid_search_align", raw, "-j", 3, "--frames", 32)
    assert "Running 54 parameter combinations using 3 threads" in out
    ratios = [float(x) for x in re.findall(r"ratio=([0-9.eE+-]+)  elapsed", out)]
    assert len(ratios) == 54 and "[skipped]" not in out          # both halves of the grid run: phase correlation is built
    best = float(re.search(r"Best params: .* ratio=([0-9.eE+-]+)", out).group(1))
    assert best == min(ratios) and best < 1.0
    # one combination re-done through the python binding gives the ratio the tool printed
    m = re.search(r"PC=0 thr=0.02 frac=0.8 maxDisp=10  outJit=([0-9.eE+-]+)  ratio=([0-9.eE+-]+)", out)
    stab = gpu_vs.Stabilizer(device=0, enable_smoother=0, lag=1, smoother_memory=0)
    outs = np.stack([o for o in (stab.process(f) for f in frames[:32]) if o is not None])
    j_out, j_in = jitter_of(gpu_vs, outs), jitter_of(gpu_vs, frames[:32])
    assert abs(float(m.group(1)) - j_out) <= 1e-4 * j_out + 1e-6
    assert abs(float(m.group(2)) - j_out / j_in) <= 1e-4


def test_grid_search_device_split_reproduces_the_single_device_search(gpu_vs, clip):
    """--devices a,b,...: the C++ many-clip split (worker t -> device slot t mod G, the clip uploaded once per slot, handles per
    worker: the independence model of grid_search_align.cpp:159-210).  The one-GPU box rehearses it with `--devices 0,0` -- two
    slots with their own clip copies and workers; every combination's ratio must equal the single-device search's, bit for bit
    (same frames, same handles' arithmetic, no state shared between workers), whichever worker got which combination."""
    d, frames, raw = clip
    one = run("vs_grid_search_align", raw, "-j", 2, "--frames", 24, "--dump-ratios")
    two = run("vs_grid_search_align", raw, "-j", 4, "--frames", 24, "--devices", "0,0", "--dump-ratios")
    r1 = re.findall(r"^RATIO (\d+) (\S+)$", one, re.M)
    r2 = re.findall(r"^RATIO (\d+) (\S+)$", two, re.M)
    assert len(r1) == 54 and r1 == r2
    assert re.search(r"Best params: .*", one).group(0) == re.search(r"Best params: .*", two).group(0)
    assert "Device split: 2 slots, 4 workers" in two and "Device split" not in one
    per_slot = [int(x) for x in re.findall(r"^  slot \d+ \(device 0\): (\d+) combinations", two, re.M)]
    assert len(per_slot) == 2 and sum(per_slot) == 54 and min(per_slot) >= 1
    out = subprocess.run([os.path.join(BIN, "vs_grid_search_align"), str(raw), "--devices", "0,7"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "No HIP device 7" in out.stderr


def test_grid_search_smoother_quick(gpu_vs, clip):
    d, frames, raw = clip
    out = run("vs_grid_search_smoother", raw, "-j", 2, "--quick")
    assert "Evaluating 48 parameter combinations using 2 threads" in out
    ratios = [float(x) for x in re.findall(r"ratio=([0-9.eE+-]+)  elapsed", out)]
    assert len(ratios) == 48
    assert float(re.search(r"jitter ratio    = ([0-9.eE+-]+)", out).group(1)) == min(ratios)
    for key in ("lag", "smoother_memory", "lambda", "min_disp", "max_disp", "min_decay", "max_decay"):
        assert re.search(r"^  %s\s+= " % key, out, re.M), key


def test_video_test_10bit_y4m(gpu_vs, tmp_path):
    """a 10-bit clip goes through the 16-bit frame path end to end: C444p10 in, C444p10 out, lag frames of latency, and the
    output is what the library's stabilizer returns for the frames the reader decodes"""
    from video_stabilizer_amd import synth
    w, h, n = 256, 192, 14
    frames, _ = synth.make_clip(w, h, n, seed=9, channels=3, bits=10)
    b, g, r = (frames[..., c].astype(np.int64) for c in range(3))
    y = np.clip(((66 * r + 129 * g + 25 * b + 128) >> 8) + 64, 0, 1023)
    cb = np.clip(((-38 * r - 74 * g + 112 * b + 128) >> 8) + 512, 0, 1023)
    cr = np.clip(((112 * r - 94 * g - 18 * b + 128) >> 8) + 512, 0, 1023)
    d = tmp_path / "in"
    d.mkdir()
    with open(d / "deep.y4m", "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F25:1 Ip A1:1 C444p10\n" % (w, h))
        for k in range(n):
            f.write(b"FRAME\n")
            for p in (y[k], cb[k], cr[k]):
                f.write(p.astype("<u2").tobytes())
    out = run("vs_video_test", d, tmp_path / "out")
    assert "Finished processing 14 frames (4 written" in out
    raw = open(tmp_path / "out" / "processed_deep.y4m", "rb").read()
    header = raw[:raw.index(b"\n")].decode()
    assert " W256 " in header and " H192 " in header and header.endswith("C444p10") and " F25:1 " in header
    fsz = w * h * 3 * 2
    assert len(raw) == len(header) + 1 + 4 * (6 + fsz)
    # decode twin (10-bit, 4:4:4) of the input, stabilized by the library, encoded again == the file's planes
    c_, d_, e_ = y - 64, cb - 512, cr - 512
    dec = np.stack([np.clip((298 * c_ + 516 * d_ + 128) >> 8, 0, 1023), np.clip((298 * c_ - 100 * d_ - 208 * e_ + 128) >> 8, 0, 1023),
                    np.clip((298 * c_ + 409 * e_ + 128) >> 8, 0, 1023)], -1).astype(np.uint16)
    stab = gpu_vs.Stabilizer(device=0, crop_pixels=0)
    want = np.stack([o for o in (stab.process(f) for f in dec) if o is not None]).astype(np.int64)
    want = np.minimum(want, 1023)                            # the writer clamps to the clip's depth
    wb, wg, wr = want[..., 0], want[..., 1], want[..., 2]
    wy = np.clip(((66 * wr + 129 * wg + 25 * wb + 128) >> 8) + 64, 0, 1023)
    pos = len(header) + 1
    for k in range(4):
        assert raw[pos:pos + 6] == b"FRAME\n"
        got_y = np.frombuffer(raw, "<u2", w * h, pos + 6).reshape(h, w)
        assert np.array_equal(got_y, wy[k]), k
        pos += 6 + fsz


def test_latency_harness_aligns_every_frame_of_its_clip(gpu_vs, clip):
    """apps/vs_latency: the reference's one-frame-per-call pattern from C++ (what profiles/r02_latency_cpp.txt is measured with);
    here only that it runs, aligns its seeded clip and reports a sane iteration count"""
    import json
    for args in ((640, 360, 8, 128), (1920, 1080, 4, 256)):
        out = json.loads(run("vs_latency", *args).strip().splitlines()[-1])
        assert out["w"] == args[0] and out["frames"] == args[2] and out["aligned"] == args[2] - 1
        assert 3 <= out["gn_iterations_per_frame"] <= 60 and out["ms_per_call"] > 0


def test_many_clips_cpp_harness_splits_clips_over_device_slots(gpu_vs, clip):
    """apps/vs_many_clips: BASELINE configs[3] with the host side in C++ -- one thread per device slot, clip i -> slot i mod G, no
    exchange, per-slot seconds.  Two slots on the one GPU here; the frames aligned must not depend on the split."""
    import json
    raw = run("vs_many_clips", "--clips", 5, "--frames", 12, "--size", "640x360", "--steps", 2, "--min-width", 64)
    assert raw.strip().startswith("{") and len(raw.strip().splitlines()) == 1    # one JSON line, no library banners on stdout
    one = json.loads(raw.strip().splitlines()[-1])
    two = json.loads(run("vs_many_clips", "--clips", 5, "--frames", 12, "--size", "640x360", "--steps", 2, "--min-width", 64, "--devices", "0,0").strip().splitlines()[-1])
    assert one["devices"] == [0] and one["per_slot_clips"] == [5] and len(one["per_slot_seconds"]) == 1
    assert two["devices"] == [0, 0] and two["per_slot_clips"] == [3, 2] and len(two["per_slot_seconds"]) == 2
    assert one["aligned_per_step"] == two["aligned_per_step"] == 5 * 11          # every frame but each clip's first
    # the report scalars travel over RCCL (one communicator per GPU) and equal the host-side sums; a device listed twice cannot have
    # two RCCL ranks: the harness says so and aggregates on the host
    assert one["aggregate"].startswith("rccl") and one["aggregate_note"] == ""
    assert two["aggregate"] == "host-side sums" and "more than once" in two["aggregate_note"]
    for j in (one, two):
        assert j["value"] > 0 and j["scaling"] == "strong" and j["seconds"] == max(j["per_slot_seconds"]) and j["warp"] == "lanczos2 separable"
        assert j["solver"] == "shared"
    # the reference's own per-frame warp (cv::warpAffine's fixed-point bilinear, constant border) in the same harness, either solver build: the frames
    # aligned do not depend on the warp or the build
    for solver in ("shared", "exclusive"):
        cv = json.loads(run("vs_many_clips", "--clips", 5, "--frames", 12, "--size", "640x360", "--steps", 2, "--min-width", 64, "--devices", "0,0",
                            "--warp-mode", "cv", "--solver", solver).strip().splitlines()[-1])
        assert cv["aligned_per_step"] == 5 * 11 and cv["solver"] == solver and cv["warp"].startswith("cv::warpAffine") and cv["value"] > 0
    # eight slots -- the shape of the driver's 8-GPU run -- on the one device: BASELINE configs[3]'s 64 clips, 8 per slot, eight per-slot
    # seconds, one JSON line; eight RCCL ranks cannot share a GPU, so the report falls back to the host-side sums and says so
    raw8 = run("vs_many_clips", "--clips", 64, "--frames", 3, "--size", "640x360", "--steps", 2, "--min-width", 64, "--devices", "0,0,0,0,0,0,0,0")
    assert len(raw8.strip().splitlines()) == 1
    eight = json.loads(raw8.strip())
    assert eight["devices"] == [0] * 8 and eight["per_slot_clips"] == [8] * 8 and len(eight["per_slot_seconds"]) == 8
    assert eight["aligned_per_step"] == 64 * 2 and eight["aggregate"] == "host-side sums" and "more than once" in eight["aggregate_note"]
    # a slot that fails ends the program with an error naming it, whatever the other seven did
    bad = subprocess.run([os.path.join(BIN, "vs_many_clips"), "--clips", "64", "--frames", "3", "--size", "640x360", "--steps", "1", "--min-width", "64",
                          "--devices", "0,0,0,0,0,0,0,0"], capture_output=True, text=True, timeout=300, env=dict(os.environ, VS_MANY_CLIPS_TEST_FAIL_SLOT="5"))
    assert bad.returncode == 1 and "slot 5 fails on purpose" in bad.stderr and "{" not in bad.stdout
    out = subprocess.run([os.path.join(BIN, "vs_many_clips"), "--devices", "0,9"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "no HIP device 9" in out.stderr
