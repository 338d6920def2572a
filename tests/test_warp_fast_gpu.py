"""VS_WARP_LANCZOS2_FAST = the contracted form of the Lanczos2 sampler, against its CPU twin -- bit for bit.

The reference compiles its generators for a target with FMA and without strict_float (CMakeLists.txt:151; no `strict_float`
in the tree), so on the reference's own machine LLVM may fuse a multiply into the add that consumes it.  The oracle's
VSO_WARP_LANCZOS2_CONTRACTED restates generators.cpp:31-47 / :684-697 with exactly those fusions (std::fmaf) and is pinned by
a literal restatement with an exact rational fma (tests/test_oracle_known_answers.py).  The product's fast mode computes the
same sequence of roundings on the GPU: every comparison below is np.array_equal -- float output (generic kernel) and
integer outputs (tuned c3 kernels, LDS path and global path), 8- and 10-bit, both borders.  No tolerance anywhere.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRANSFORMS = [(0.004, -0.003, 2.25, -1.5), (-0.01, 0.02, -7.75, 3.125), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 3.0, -2.0), (0.0007, 0.0019, 0.5, 0.5)]


@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("bits", [8, 10])
def test_fast_mode_float_output_equals_the_contracted_twin(gpu_vs, oracle, bits, border):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 200, 1, seed=11, channels=3, bits=bits)
    src = frames[0]
    differs = 0
    for tr in TRANSFORMS:
        tg, to = gpu_vs.Transform.of(*tr), oracle.Transform.of(*tr)
        fast = gpu_vs.bgr_image_warp(src, tg, mode=gpu_vs.WARP_LANCZOS2_FAST, border=border, f32=True)
        twin = oracle.bgr_image_warp(src, to, oracle.WARP_LANCZOS2_CONTRACTED, border=border, f32=True)
        assert np.array_equal(fast, twin), (tr, float(np.abs(fast - twin).max()))
        exact = gpu_vs.bgr_image_warp(src, tg, mode=gpu_vs.WARP_LANCZOS2, border=border, f32=True)
        assert np.array_equal(exact, oracle.bgr_image_warp(src, to, border=border, f32=True))   # the exact mode IS the un-contracted oracle
        differs += int(not np.array_equal(exact, fast))
    assert differs > 0          # the two modes are different functions: the twin tests something


@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("dtype,hi,bits", [(np.uint8, 255, 8), (np.uint16, 1023, 10)])
def test_fast_mode_integer_output_equals_the_contracted_twin(gpu_vs, oracle, dtype, hi, bits, border):
    """Tuned c3 kernels (u8 / u16): identical to the twin; and never more than 1 LSB from the un-contracted mode."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 360, 3, seed=5, channels=3, bits=bits)
    ts = [gpu_vs.Transform.of(*tr) for tr in TRANSFORMS[:3]]
    fast = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_LANCZOS2_FAST, border=border, max_value=hi)
    exact = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_LANCZOS2, border=border, max_value=hi)
    for i in range(3):
        to = oracle.Transform.of(*ts[i].tup())
        assert np.array_equal(fast[i], oracle.bgr_image_warp(frames[i], to, oracle.WARP_LANCZOS2_CONTRACTED, border=border, max_value=hi))
        assert np.array_equal(exact[i], oracle.bgr_image_warp(frames[i], to, border=border, max_value=hi))
    d = np.abs(fast.astype(np.int64) - exact.astype(np.int64))
    print("bits", bits, "border", border, "identical to the un-contracted mode:", float((d == 0).mean()), "max diff", int(d.max()))
    assert d.max() <= 1


def test_fast_mode_ragged_sizes_windows_and_unaligned_rows(gpu_vs, oracle):
    """sizes that are not multiples of the 64x16 tile, of 4 pixels or of 4 bytes per row: border tiles, byte-wise stores"""
    rng = np.random.default_rng(21)
    for (h, w) in [(17, 65), (33, 130), (16, 64), (5, 7), (70, 201)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for tr in TRANSFORMS[:2]:
            for border in (0, 1):
                got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_LANCZOS2_FAST, border=border)
                want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_LANCZOS2_CONTRACTED, border=border)
                assert np.array_equal(got, want), (h, w, tr, border)


def test_fast_mode_other_layouts_use_the_same_arithmetic(gpu_vs, oracle):
    """1-channel frames have no tuned kernel: the generic kernel serves them with the same contracted arithmetic."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(256, 160, 1, seed=3, channels=3)
    gray = np.ascontiguousarray(frames[0][..., :1])
    bgr = np.ascontiguousarray(np.repeat(gray, 3, axis=2))
    t = gpu_vs.Transform.of(0.003, -0.002, 1.25, -0.75)
    a = gpu_vs.bgr_image_warp(gray, t, mode=gpu_vs.WARP_LANCZOS2_FAST)
    b = gpu_vs.bgr_image_warp(bgr, t, mode=gpu_vs.WARP_LANCZOS2_FAST)
    assert np.array_equal(a[..., 0], b[..., 0]) and np.array_equal(b[..., 0], b[..., 2])
    assert np.array_equal(a, oracle.bgr_image_warp(gray, oracle.Transform.of(*t.tup()), oracle.WARP_LANCZOS2_CONTRACTED))


def test_fast_mode_large_rotation_takes_the_global_path(gpu_vs, oracle):
    """Footprints that do not fit the LDS window run the per-pixel path of the same kernel, same arithmetic."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 1, seed=9, channels=3)
    t = gpu_vs.Transform.of(-0.2, 0.6, 4.0, -3.0)
    fast = gpu_vs.bgr_image_warp(frames[0], t, mode=gpu_vs.WARP_LANCZOS2_FAST)
    assert np.array_equal(fast, oracle.bgr_image_warp(frames[0], oracle.Transform.of(*t.tup()), oracle.WARP_LANCZOS2_CONTRACTED))


def test_fast_mode_4k_frame_equals_the_contracted_twin(gpu_vs, oracle):
    """BASELINE configs[2] frame size, one frame (the oracle needs a few seconds for it with its rows in parallel)."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(3840, 2160, 1, seed=2, channels=3)
    t = (0.0012, -0.0017, 3.3, -2.7)
    oracle.set_threads(8)
    try:
        want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*t), oracle.WARP_LANCZOS2_CONTRACTED)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_LANCZOS2_FAST), want)
