"""The gate that decides what bench.py's `value` may be ("parity gates run with every benchmark"): the PRODUCT's
contracted sampler (VS_WARP_LANCZOS2_FAST, the tuned c3 kernels) against the UN-CONTRACTED oracle (VSO_WARP_LANCZOS2 = the
reference's written rounding order) -- not against its own twin.

  u8 / u16 (10-bit):  max |d| <= 1 LSB  and  identical fraction >= 0.9999     asserted, 1080p parity frames + one 4K frame
  f32:                as close to the real-arithmetic value as the reference's order is (the formula of 8(d) is structurally
                      out of reach: tests/test_warp_gate_cpu.py, oracle/gate.py)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRANSFORMS = [(0.0009, -0.0016, 3.7, -2.9), (-0.0007, 0.0012, -1.25, 3.5), (0.004, -0.003, 2.25, -1.5), (0.0, 0.0, 3.25, -1.75)]


@pytest.fixture(scope="module")
def gate(oracle):
    from oracle import gate as G
    oracle.set_threads(8)
    yield G
    oracle.set_threads(1)


@pytest.mark.parametrize("bits,hi", [(8, 255), (10, 1023)])
def test_contracted_gpu_warp_vs_uncontracted_oracle_1080p(gpu_vs, oracle, gate, bits, hi):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(1920, 1080, 4, seed=1, channels=3, bits=bits)
    ts = [gpu_vs.Transform.of(*tr) for tr in TRANSFORMS]
    fast = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_LANCZOS2_FAST, max_value=hi)
    worst = 1.0
    for i in range(4):
        want = oracle.bgr_image_warp(frames[i], oracle.Transform.of(*TRANSFORMS[i]), oracle.WARP_LANCZOS2, max_value=hi)
        ok, info = gate.integer_gate(fast[i], want)
        assert ok, (bits, i, info)
        worst = min(worst, info["identical_fraction"])
    print("%d-bit 1080p: contracted GPU vs un-contracted oracle, least identical fraction %.6f" % (bits, worst))


def test_contracted_gpu_warp_vs_uncontracted_oracle_4k(gpu_vs, oracle, gate):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(3840, 2160, 1, seed=2, channels=3)
    tr = (0.0012, -0.0017, 3.3, -2.7)
    want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr), oracle.WARP_LANCZOS2)
    got = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_LANCZOS2_FAST)
    ok, info = gate.integer_gate(got, want)
    print("4K:", info)
    assert ok, info


def test_contracted_gpu_float_output_is_as_close_to_real_arithmetic_as_the_reference_order(gpu_vs, oracle, gate):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 360, 1, seed=5, channels=3)
    for tr in TRANSFORMS[:3]:
        exact = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr), oracle.WARP_LANCZOS2, f32=True).astype(np.float64)
        fast = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_LANCZOS2_FAST, f32=True).astype(np.float64)
        real = gate.lanczos_real(frames[0], tr)
        de, dc = np.abs(exact - real), np.abs(fast - real)
        assert dc.max() <= 1.25 * de.max() + 1e-9 and np.sqrt((dc ** 2).mean()) <= 1.1 * np.sqrt((de ** 2).mean()) + 1e-12
        assert np.abs(exact - fast).max() <= 1.5e-3


def test_exact_half_pixel_translation_splits_only_on_rounding_ties(gpu_vs, oracle, gate):
    """The one family of transforms where the identical fraction drops below 99.99 %: a pure translation by EXACTLY half a pixel in
    both axes.  The tap weights are then symmetric, so for ~0.4 % of the samples the real-arithmetic value of the formula is exactly
    k + 0.5 -- a tie of the store rule (round half up) -- and which side an fp32 evaluation lands on is decided by its last bit.
    Still never more than 1 LSB, and EVERY sample on which the two forms differ is such a tie: neither output is the wrong one.
    (Measured transforms are never exact binary fractions; bench.py runs the gate on the transforms the timed step used.)"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(1920, 1080, 1, seed=4, channels=3)
    tr = (0.0, 0.0, 0.5, 0.5)
    got = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_LANCZOS2_FAST)
    want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr), oracle.WARP_LANCZOS2)
    ok, info = gate.integer_gate(got, want)
    print("exact half-pixel shift:", info)
    assert info["max_abs_diff_lsb"] <= 1 and info["identical_fraction"] >= 0.995
    real = gate.lanczos_real(frames[0], tr)
    differs = got != want
    assert differs.any() and (np.abs(real[differs] - np.floor(real[differs]) - 0.5) < 1e-3).all()
