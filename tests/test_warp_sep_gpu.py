"""VS_WARP_LANCZOS2_SEP = the separable member of the Lanczos2 sampler family, against its CPU twin -- bit for bit -- and through
SURVEY 8(d)'s integer gate against the UN-contracted order (the reference's written sequence of roundings).

The twin (oracle/vs_oracle.cpp lanczos_sample_separable) is std::fmaf + one IEEE divide: the same function on every machine.  The
product computes the same roundings on the GPU (generic kernel: float and integer outputs, any layout; tuned c3 kernels: LDS path,
global path, 8- and 16-bit containers, both borders).  Every twin comparison is np.array_equal.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TRANSFORMS = [(0.004, -0.003, 2.25, -1.5), (-0.01, 0.02, -7.75, 3.125), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 3.0, -2.0), (0.0007, 0.0019, 0.5, 0.5)]


@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("bits", [8, 10])
def test_sep_mode_float_output_equals_the_separable_twin(gpu_vs, oracle, bits, border):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 200, 1, seed=11, channels=3, bits=bits)
    src = frames[0]
    differs = 0
    for tr in TRANSFORMS:
        tg, to = gpu_vs.Transform.of(*tr), oracle.Transform.of(*tr)
        sep = gpu_vs.bgr_image_warp(src, tg, mode=gpu_vs.WARP_LANCZOS2_SEP, border=border, f32=True)
        twin = oracle.bgr_image_warp(src, to, oracle.WARP_LANCZOS2_SEPARABLE, border=border, f32=True)
        assert np.array_equal(sep, twin), (tr, float(np.abs(sep - twin).max()))
        fast = gpu_vs.bgr_image_warp(src, tg, mode=gpu_vs.WARP_LANCZOS2_FAST, border=border, f32=True)
        differs += int(not np.array_equal(sep, fast))
    assert differs > 0          # a different function from the contracted form: the twin tests something


@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("dtype,hi,bits", [(np.uint8, 255, 8), (np.uint16, 1023, 10), (np.uint16, 65535, 16)])
def test_sep_mode_integer_output_equals_the_separable_twin_and_passes_the_gate(gpu_vs, oracle, dtype, hi, bits, border):
    """Tuned c3 kernels (u8 / u16): identical to the twin; SURVEY 8(d)'s integer gate against the UN-contracted mode."""
    from oracle import gate as G
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 360, 3, seed=5, channels=3, bits=bits)
    ts = [gpu_vs.Transform.of(*tr) for tr in TRANSFORMS[:3]]
    sep = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_LANCZOS2_SEP, border=border, max_value=hi)
    exact = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_LANCZOS2, border=border, max_value=hi)
    for i in range(3):
        to = oracle.Transform.of(*ts[i].tup())
        assert np.array_equal(sep[i], oracle.bgr_image_warp(frames[i], to, oracle.WARP_LANCZOS2_SEPARABLE, border=border, max_value=hi)), i
    ok, info = G.integer_gate(sep, exact)
    print("bits", bits, "border", border, "separable vs un-contracted:", info)
    if bits <= 10:
        assert ok, info
    else:
        assert info["max_abs_diff_lsb"] <= 64          # 16-bit samples: 2^-8 of an 8-bit step is the fp32 sampler's own resolution


def test_sep_mode_ragged_sizes_windows_and_unaligned_rows(gpu_vs, oracle):
    """sizes that are not multiples of the 64x16 tile, of 4 pixels or of 4 bytes per row: border tiles, byte-wise stores"""
    rng = np.random.default_rng(21)
    for (h, w) in [(17, 65), (33, 130), (16, 64), (5, 7), (70, 201), (1, 1)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for tr in TRANSFORMS[:2]:
            for border in (0, 1):
                got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_LANCZOS2_SEP, border=border)
                want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_LANCZOS2_SEPARABLE, border=border)
                assert np.array_equal(got, want), (h, w, tr, border)


def test_sep_mode_fractions_on_the_select_boundaries(gpu_vs, oracle):
    """Integer and near-integer positions: fractions exactly 0 (tap 4's argument is exactly 2) and fractions that round up to 1 (a
    position a hair below an integer: tap 1's argument is exactly -2) take the kernels' select path, every other block the
    select-free one -- both must be the twin.  The transforms put such positions on some tiles and not on others."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 360, 1, seed=13, channels=3)
    src = frames[0]
    cases = [(0.0, 0.0, 0.0, 0.0), (0.0, 0.0, -1e-9, -1e-9), (0.0, 0.0, 2.0, 0.3), (0.0, 0.0, 0.3, -3.0), (0.001, 0.0, 0.0, 0.0),
             (0.0, 0.001, 0.0, 0.0), (2.0 ** -10, 0.0, 0.5, 0.25), (0.0, 2.0 ** -9, 0.125, 0.0)]
    for mode, omode in ((gpu_vs.WARP_LANCZOS2_SEP, oracle.WARP_LANCZOS2_SEPARABLE), (gpu_vs.WARP_LANCZOS2_FAST, oracle.WARP_LANCZOS2_CONTRACTED)):
        for tr in cases:
            got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=mode)
            want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), omode)
            assert np.array_equal(got, want), (mode, tr, int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_sep_mode_other_layouts_and_the_global_path(gpu_vs, oracle):
    """1-channel frames have no tuned kernel (generic kernel, same arithmetic); large rotations take the tuned kernel's global path."""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 1, seed=3, channels=3)
    gray = np.ascontiguousarray(frames[0][..., :1])
    t = (0.003, -0.002, 1.25, -0.75)
    a = gpu_vs.bgr_image_warp(gray, gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_LANCZOS2_SEP)
    assert np.array_equal(a, oracle.bgr_image_warp(gray, oracle.Transform.of(*t), oracle.WARP_LANCZOS2_SEPARABLE))
    t = (-0.2, 0.6, 4.0, -3.0)
    b = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_LANCZOS2_SEP)
    assert np.array_equal(b, oracle.bgr_image_warp(frames[0], oracle.Transform.of(*t), oracle.WARP_LANCZOS2_SEPARABLE))


def test_sep_mode_4k_frame_equals_the_separable_twin_and_passes_the_gate(gpu_vs, oracle):
    """BASELINE configs[2] frame size, one frame, 8- and 10-bit: the twin bit for bit, the gate against the un-contracted order."""
    from oracle import gate as G
    from video_stabilizer_amd import synth
    t = (0.0012, -0.0017, 3.3, -2.7)
    oracle.set_threads(8)
    try:
        for bits, hi in ((8, 255), (10, 1023)):
            frames, _ = synth.make_clip(3840, 2160, 1, seed=2, channels=3, bits=bits)
            got = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_LANCZOS2_SEP, max_value=hi)
            assert np.array_equal(got, oracle.bgr_image_warp(frames[0], oracle.Transform.of(*t), oracle.WARP_LANCZOS2_SEPARABLE, max_value=hi))
            exact = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_LANCZOS2, max_value=hi)
            ok, info = G.integer_gate(got, exact)
            print("4K", bits, "bit separable vs un-contracted:", info)
            assert ok, (bits, info)
    finally:
        oracle.set_threads(1)


def test_compact_six_wave_instantiation_equals_the_standard_one():
    """round 6: the contracted / separable forms run a COMPACT instantiation (20-row window, six waves per SIMD) when the host-side extents say every tile's footprint
    spans under 16 source rows, the standard one (24 rows) otherwise or with VS_WARP_COMPACT=0 (read once per process: child processes).  Same arithmetic, same bits:
    transforms either side of the threshold (rotation ~0.9 degrees), 8- and 10-bit, both borders, a batch that mixes fitting and non-fitting frames."""
    import hashlib
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from video_stabilizer_amd import capi\n"
            "rng = np.random.default_rng(9)\n"
            "h = hashlib.sha256()\n"
            "for bits in (8, 10):\n"
            "    mv = 255 if bits == 8 else 1023\n"
            "    src = rng.integers(0, mv + 1, (3, 150, 333, 3)).astype(np.uint8 if bits == 8 else np.uint16)\n"
            "    for mode in (capi.WARP_LANCZOS2_SEP, capi.WARP_LANCZOS2_FAST):\n"
            "        for border in (0, 1):\n"
            "            for trs in ([(0.002, -0.0015, 3.3, -2.7)] * 3, [(0.001, 0.014, 1.0, 2.0), (0.0, 0.0155, -3.0, 0.5), (0.0, 0.0165, 0.25, 0.75)],\n"
            "                        [(0.03, 0.02, 5.0, -4.0), (0.0, 0.0, 0.5, 0.5), (-0.05, 0.1, 0.0, 0.0)]):\n"
            "                out = capi.bgr_image_warp_batch(src, [capi.Transform.of(*t) for t in trs], mode, border, max_value=mv)\n"
            "                h.update(out.tobytes())\n"
            "print('DIGEST', h.hexdigest())\n") % root
    digests = []
    for env in ({}, {"VS_WARP_COMPACT": "0"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(out.stdout.strip().split("DIGEST")[-1].strip())
    assert digests[0] == digests[1] and len(digests[0]) == 64
