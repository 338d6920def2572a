"""VS_WARP_BILINEAR_CV = cv::warpAffine(INTER_LINEAR) as the reference's stabilizer calls it (stabilizer.cpp:97-99 -> imgproc.cpp:446-484),
OpenCV 4.x's fixed-point path, against its CPU twin (oracle/vs_oracle.cpp cv_warp_impl) -- integer work: np.array_equal everywhere, no
tolerance.  The twin itself is held against an independent numpy coding of the same published algorithm in
tests/test_bilinear_vs_opencv_fixed_point.py; both are the builder's reading of OpenCV ("parity unpinned (OpenCV version)").
In this mode the transform argument is the FORWARD map handed to warpBySimilarityTransform (cv::warpAffine inverts it itself)."""
import numpy as np
import pytest

from _diff import same

pytestmark = pytest.mark.gpu

TRANSFORMS = [(0.004, -0.003, 2.25, -1.5), (-0.01, 0.02, -7.75, 3.125), (0.0, 0.0, 0.0, 0.0), (0.0, 0.0, 3.0, -2.0), (0.0007, 0.0019, 0.5, 0.5),
              (0.0, 0.0, 1.0 / 64, -1.0 / 64), (-0.002, 0.0015, -6.4, 3.3)]


@pytest.mark.parametrize("border", [0, 1])
def test_cv_mode_8bit_bgr_equals_the_twin(gpu_vs, oracle, border):
    """the tuned kernel (byte tile in LDS, v_dot2_u32_u16 taps): whole frames, a batch, both borders"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 360, len(TRANSFORMS), seed=5, channels=3)
    ts = [gpu_vs.Transform.of(*tr) for tr in TRANSFORMS]
    got = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_BILINEAR_CV, border=border)
    for i, tr in enumerate(TRANSFORMS):
        want = oracle.bgr_image_warp(frames[i], oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border)
        assert np.array_equal(got[i], want), (tr, border, int(np.abs(got[i].astype(int) - want.astype(int)).max()))
        one = gpu_vs.bgr_image_warp(frames[i], ts[i], mode=gpu_vs.WARP_BILINEAR_CV, border=border)
        assert np.array_equal(one, want), (tr, border)


def test_cv_mode_is_the_forward_map_cv_warp_affine_inverts(gpu_vs, oracle):
    """warpBySimilarityTransform(src, T) moves content BY T (cv::warpAffine without WARP_INVERSE_MAP): a pure translation by (+5, +7)
    puts source pixel (x, y) at (x + 5, y + 7)"""
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (90, 130, 3), dtype=np.uint8)
    out = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(0.0, 0.0, 5.0, 7.0), mode=gpu_vs.WARP_BILINEAR_CV, border=gpu_vs.BORDER_CONSTANT)
    assert np.array_equal(out[7:, 5:], src[:-7, :-5]) and not out[:7].any() and not out[:, :5].any()


def test_cv_mode_every_fraction_pair_with_extreme_samples(gpu_vs, oracle):
    """The tuned 8-bit sampler works with 16-bit weights 64 a b, its top-left weight saturated to 65535 where it would be 65536 (fx = fy = 0):
    translations by every pair of 1/32-pixel fractions (1024 frames: each (fx, fy) is the fraction of EVERY interior sample of its frame) over an
    image of saturated and near-saturated bytes, against the twin."""
    rng = np.random.default_rng(77)
    src = rng.choice(np.array([0, 1, 127, 128, 254, 255], np.uint8), size=(40, 72, 3))
    src[::3, ::2] = 255
    trs = [(0.0, 0.0, 1.0 + fx / 32.0, -1.0 - fy / 32.0) for fx in range(32) for fy in range(32)]
    frames = np.ascontiguousarray(np.broadcast_to(src, (len(trs),) + src.shape))
    got = gpu_vs.bgr_image_warp_batch(frames, [gpu_vs.Transform.of(*t) for t in trs], mode=gpu_vs.WARP_BILINEAR_CV, border=gpu_vs.BORDER_CONSTANT)
    for i, tr in enumerate(trs):
        want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=oracle.BORDER_CONSTANT)
        assert np.array_equal(got[i], want), tr


def test_cv_mode_ragged_sizes_unaligned_rows_windows(gpu_vs, oracle):
    rng = np.random.default_rng(21)
    for (h, w) in [(17, 65), (33, 130), (32, 64), (5, 7), (70, 201), (1, 1), (40, 63)]:
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for tr in TRANSFORMS[:2] + [(0.0, 0.0, 0.25, -0.75)]:
            for border in (0, 1):
                got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=border)
                want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border)
                assert np.array_equal(got, want), (h, w, tr, border)
    # an output window equals the same rows / columns cut out of the whole warp (the stabilizer's crop)
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(480, 270, 2, seed=9, channels=3)
    ts = [gpu_vs.Transform.of(*TRANSFORMS[0]), gpu_vs.Transform.of(*TRANSFORMS[6])]
    full = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_BILINEAR_CV, border=1)
    for roi in [(32, 32, 416, 206), (1, 3, 77, 40), (100, 50, 64, 32)]:
        win = gpu_vs.bgr_image_warp_roi_batch(frames, ts, roi, mode=gpu_vs.WARP_BILINEAR_CV, border=1)
        x, y, rw, rh = roi
        assert np.array_equal(win, full[:, y:y + rh, x:x + rw]), roi


def test_cv_mode_large_rotation_and_zoom_take_the_global_path(gpu_vs, oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 1, seed=9, channels=3)
    for tr in [(-0.2, 0.6, 4.0, -3.0), (1.5, 0.0, 0.0, 0.0), (-0.7, 0.1, 30.0, 10.0), (0.0, 0.0, 40000.0, 0.0), (0.0, 0.0, -1e7, 3e6)]:
        for border in (0, 1):
            got = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=border)
            want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border)
            assert np.array_equal(got, want), (tr, border)


@pytest.mark.parametrize("channels", [1, 2, 4])
def test_cv_mode_other_channel_counts(gpu_vs, oracle, channels):
    rng = np.random.default_rng(channels)
    src = rng.integers(0, 256, (120, 200, channels), dtype=np.uint8)
    for tr in TRANSFORMS[:3]:
        got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=1)
        assert np.array_equal(got, oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=1)), (channels, tr)


@pytest.mark.parametrize("bits,hi", [(10, 1023), (16, 65535)])
def test_cv_mode_16bit_containers(gpu_vs, oracle, bits, hi):
    """remapBilinear<Cast<float, ushort>>: float weights (exact), float products and sums left to right, cvRound"""
    from video_stabilizer_amd import synth
    f10, _ = synth.make_clip(320, 200, 1, seed=11, channels=3, bits=10)
    src = f10[0] if bits == 10 else (f10[0].astype(np.uint32) * 64 + 37).astype(np.uint16)
    for tr in TRANSFORMS[:4]:
        for border in (0, 1):
            got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=border, max_value=hi)
            want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border, max_value=hi)
            assert np.array_equal(got, want), (bits, tr, border)


def test_cv_mode_16bit_containers_ragged_sizes_mixed_depth_tiles_and_windows(gpu_vs, oracle):
    """the tuned word-tile kernel: sizes that are not multiples of its 64 x 32 tile or of a pixel pair, unaligned rows, a frame whose samples stay
    below 2^14 everywhere but in one corner (tiles there evaluate OpenCV's float expression as written, the others its exact integer form: both must
    be the twin), 12-bit content, an output window"""
    rng = np.random.default_rng(5)
    for (h, w) in [(17, 65), (33, 130), (32, 64), (5, 7), (70, 201), (1, 1)]:
        src = rng.integers(0, 4096, (h, w, 3), dtype=np.uint16)
        for tr in TRANSFORMS[:2] + [(0.0, 0.0, 0.25, -0.75)]:
            for border in (0, 1):
                got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=border, max_value=4095)
                want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border, max_value=4095)
                assert np.array_equal(got, want), (h, w, tr, border)
    src = rng.integers(0, 16384, (200, 300, 3), dtype=np.uint16)
    src[150:, 200:] = rng.integers(0, 65536, (50, 100, 3), dtype=np.uint16)
    src[10, 10] = (16383, 16384, 65535)                       # (the boundary itself)
    for tr in TRANSFORMS[:3]:
        got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=1, max_value=65535)
        assert np.array_equal(got, oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=1, max_value=65535)), tr
    frames = np.stack([src, src[::-1].copy()])
    ts = [gpu_vs.Transform.of(*TRANSFORMS[0]), gpu_vs.Transform.of(*TRANSFORMS[6])]
    full = gpu_vs.bgr_image_warp_batch(frames, ts, mode=gpu_vs.WARP_BILINEAR_CV, border=1, max_value=65535)
    for roi in [(32, 32, 200, 100), (1, 3, 77, 40)]:
        win = gpu_vs.bgr_image_warp_roi_batch(frames, ts, roi, mode=gpu_vs.WARP_BILINEAR_CV, border=1, max_value=65535)
        x, y, rw, rh = roi
        assert np.array_equal(win, full[:, y:y + rh, x:x + rw]), roi
    # every exact half (the weights are multiples of 1/1024, so halves happen): cvRound rounds them to even in both forms
    flat = np.zeros((40, 70, 3), np.uint16)
    flat[:, ::2] = 1
    got = gpu_vs.bgr_image_warp(flat, gpu_vs.Transform.of(0.0, 0.0, 0.5, 0.0), mode=gpu_vs.WARP_BILINEAR_CV, border=gpu_vs.BORDER_CONSTANT, max_value=1023)
    assert np.array_equal(got, oracle.bgr_image_warp(flat, oracle.Transform.of(0.0, 0.0, 0.5, 0.0), oracle.WARP_BILINEAR_CV, border=oracle.BORDER_CONSTANT, max_value=1023))
    assert got.max() == 0 and flat.max() == 1                  # 0.5 -> 0 (half to even), not 1


def test_cv_mode_16bit_every_fraction_pair_clamped_and_clampless_tiles(gpu_vs, oracle):
    """the word-tile kernel has two sampler forms, chosen per tile by `seen & 0xc000c000` (does ANY staged sample reach 2^14?): below it the INTEGER form --
    OpenCV's 15-bit weights 32 a b, S = sum w v by v_dot2_u32_u16, result (S + 16383 + ((S >> 15) & 1)) >> 15 = cvRound's half-to-even, then
    min(., max_value), always -- and from 2^14 up the FLOAT expression as OpenCV writes it.  This frame keeps every tile on the integer side of the
    boundary, with stray samples up to 2^14 - 1 = 16383 in one region (above max_value: the clamp must bite) and 10-bit content in the other; every
    pair of 1/32-pixel fractions, extreme and odd samples (exact halves round to even).  The float side of the boundary: test_cv_mode_16bit_*
    above (full 16-bit content) and the mixed-tile frame below."""
    rng = np.random.default_rng(78)
    src = rng.choice(np.array([0, 1, 2, 3, 511, 512, 1021, 1022, 1023], np.uint16), size=(40, 136, 3))
    src[:, 70:] = rng.choice(np.array([0, 1, 1023, 1024, 1025, 8191, 16382, 16383], np.uint16), size=(40, 66, 3))     # above max_value, below 2^14
    trs = [(0.0, 0.0, 1.0 + fx / 32.0, -1.0 - fy / 32.0) for fx in range(32) for fy in range(32)]
    frames = np.ascontiguousarray(np.broadcast_to(src, (len(trs),) + src.shape))
    got = gpu_vs.bgr_image_warp_batch(frames, [gpu_vs.Transform.of(*t) for t in trs], mode=gpu_vs.WARP_BILINEAR_CV, border=gpu_vs.BORDER_CONSTANT, max_value=1023)
    assert got.max() == 1023
    for i, tr in enumerate(trs):
        want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=oracle.BORDER_CONSTANT, max_value=1023)
        assert np.array_equal(got[i], want), tr
    # a max_value that is not 2^k - 1 always clamps
    g = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(0.001, 0.002, 0.3, 0.7), mode=gpu_vs.WARP_BILINEAR_CV, border=0, max_value=1000)
    assert np.array_equal(g, oracle.bgr_image_warp(src, oracle.Transform.of(0.001, 0.002, 0.3, 0.7), oracle.WARP_BILINEAR_CV, border=0, max_value=1000))
    # the boundary itself: ONE sample of 16384 = 2^14 in a tile flips that tile (and only it) to the float form; both forms must agree with the twin on
    # the tiles either side of it (64 x 32 output tiles: the sample sits in the second tile column)
    edge = src.copy()
    edge[:, 70:] = np.minimum(edge[:, 70:], 16383)
    edge[5, 100, 1] = 16384
    for tr in [(0.0, 0.0, 1.0 + 7 / 32.0, -1.0 - 13 / 32.0), (0.003, -0.002, 0.4, 0.9)]:
        g = gpu_vs.bgr_image_warp(edge, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=gpu_vs.BORDER_CONSTANT, max_value=65535)
        assert same(g, oracle.bgr_image_warp(edge, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=oracle.BORDER_CONSTANT, max_value=65535)), tr


@pytest.mark.parametrize("shape", [(9000, 70), (70, 9000), (16500, 66), (66, 16500)])
def test_cv_mode_frames_beyond_8192_and_16384_pixels(gpu_vs, oracle, shape):
    """the tuned path's "fits" test bounds the fixed-point table entries: 2^24 = 16384 pixels since round 6 (2^23 before: frames beyond 8192 pixels fell
    to the per-pixel path -- still right, 2-4x slower); beyond 16384 the per-pixel path takes over inside one frame.  Both depths, both borders,
    a rotation that moves the far end of the frame by tens of pixels."""
    w, h = shape
    rng = np.random.default_rng(w + h)
    for bits in (8, 10):
        mv = 255 if bits == 8 else 1023
        src = rng.integers(0, mv + 1, (h, w, 3)).astype(np.uint8 if bits == 8 else np.uint16)
        for border, tr in ((0, (0.001, -0.002, 3.25, -1.5)), (1, (-0.0005, 0.0011, -2.75, 4.5))):
            oracle.set_threads(8)
            try:
                want = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=border, max_value=mv)
            finally:
                oracle.set_threads(1)
            got = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode=gpu_vs.WARP_BILINEAR_CV, border=border, max_value=mv)
            assert same(got, want), (shape, bits, border)


def test_cv_mode_row_pitch_of_2_to_the_24_bytes_and_more(gpu_vs, oracle):
    """the interior fills address rows with 24-bit multiplies (row x pitch): a caller's pitch of 2^24 bytes or more (allowed: the API bounds the extents,
    not the pitch) must take the rim path instead of wrapping.  Three rows 2^24 + 4 bytes apart (8-bit) and 2^23 + 2 elements apart (16-bit
    containers), device memory, an interior-sized frame."""
    import ctypes as C
    import torch
    w, h = 200, 3
    rng = np.random.default_rng(5)
    for bits, pitch in ((8, (1 << 24) + 4), (16, (1 << 23) + 2)):                 # pitch in elements
        dt, tdt, mv = (np.uint8, torch.uint8, 255) if bits == 8 else (np.uint16, torch.int16, 1023)
        img = rng.integers(0, mv + 1, (h, w, 3)).astype(dt)
        src = torch.zeros((h - 1) * pitch + 3 * w, dtype=tdt, device="cuda:0")
        for y in range(h):
            row = torch.from_numpy(img[y].reshape(-1).view(np.int16) if bits == 16 else img[y].reshape(-1)).to("cuda:0")
            src[y * pitch:y * pitch + 3 * w] = row
        dst = torch.zeros((h, 3 * w), dtype=tdt, device="cuda:0")
        tr = (0.001, -0.002, 0.3, 0.2)
        t = gpu_vs.Transform.of(*tr)
        r = gpu_vs.lib().vs_bgr_image_warp(C.c_void_p(src.data_ptr()), w, h, pitch, 3, bits, C.byref(t), gpu_vs.WARP_BILINEAR_CV, gpu_vs.BORDER_CLAMP, mv,
                                           C.c_void_p(dst.data_ptr()), 3 * w, gpu_vs.MEM_DEVICE, None)
        assert r >= 0, gpu_vs.lib().vs_last_error()
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        got = (got.view(np.uint16) if bits == 16 else got).reshape(h, w, 3)
        assert same(got, oracle.bgr_image_warp(img, oracle.Transform.of(*tr), oracle.WARP_BILINEAR_CV, border=oracle.BORDER_CLAMP, max_value=mv)), bits


def test_cv_mode_4k_frame(gpu_vs, oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(3840, 2160, 1, seed=2, channels=3)
    t = (0.0012, -0.0017, 3.3, -2.7)
    oracle.set_threads(8)
    try:
        want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*t), oracle.WARP_BILINEAR_CV, border=1)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*t), mode=gpu_vs.WARP_BILINEAR_CV, border=1), want)


def test_cv_mode_has_no_float_output_and_the_inverse_matrix_is_opencvs(gpu_vs, oracle):
    src = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(gpu_vs.VsError):
        gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(), mode=gpu_vs.WARP_BILINEAR_CV, f32=True)
    # the host-side inversion is cv::warpAffine's, operation for operation: equal to the oracle's doubles bit for bit
    rng = np.random.default_rng(1)
    for _ in range(200):
        tr = (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-50, 50), rng.uniform(-50, 50))
        w, h = int(rng.integers(8, 4000)), int(rng.integers(8, 2200))
        assert np.array_equal(gpu_vs.cv_inverse_matrix(gpu_vs.Transform.of(*tr), w, h), oracle.cv_inverse_matrix(oracle.Transform.of(*tr), w, h))


def test_stabilizer_with_the_references_own_warp(gpu_vs, oracle):
    """VideoStabilizer::processFrame with warp_mode = VS_WARP_BILINEAR_CV: the correction goes to the warp as the reference hands it to
    cv::warpAffine (stabilizer.cpp:97-99); every output frame equals the oracle's"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(480, 270, 16, seed=31, channels=3)
    g = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_BILINEAR_CV, lag=3)
    c = oracle.Stabilizer(warp_mode=oracle.WARP_BILINEAR_CV, lag=3)
    outs = 0
    for f in frames:
        a, b = g.process(f), c.process(f)
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a, b)
            outs += 1
    assert outs == 16 - 3
