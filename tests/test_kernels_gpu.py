"""GPU parity of every kernel-level C-ABI entry point against the CPU oracle, on seeded inputs.

Bars (BASELINE.json north_star / SURVEY 8d): integer, byte and index outputs bit-exact; fp32
outputs bit-exact too where the kernel keeps the reference's evaluation order (they all do);
the fp64 sparse_ica sums to 1e-12 relative (tree reduction vs the reference's serial order).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(640, 480), (322, 241), (97, 61), (1920, 1080)]


def _img(w, h, seed):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(w, h, 1, seed=seed, path=[(0.0, 0.0, 0.0, 0.0)], margin=8)
    return frames[0]


def _noise(w, h, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


@pytest.mark.parametrize("w,h", SIZES + [(8, 8), (9, 7), (130, 34), (4096, 37)])
def test_pyr_down_bit_exact(gpu_vs, oracle, w, h):
    img = _noise(w, h, 1)
    assert np.array_equal(gpu_vs.pyr_down(img), oracle.pyr_down(img))


@pytest.mark.parametrize("w,h", [(2, 2), (3, 2), (4, 4), (5, 7), (7, 5), (8, 3), (9, 9), (254, 17), (255, 18), (256, 19), (257, 33), (258, 34), (259, 35),
                                 (511, 64), (513, 66), (1027, 5), (130, 257),
                                 (245, 9), (246, 9), (247, 9), (248, 9), (249, 9), (250, 9), (251, 9), (252, 10), (253, 11), (495, 5), (496, 5), (497, 5),
                                 (499, 7), (743, 6), (744, 6), (745, 6), (6, 40), (4, 33)])
def test_pyr_down_strip_and_band_boundaries(gpu_vs, oracle, w, h):
    """the row-walking kernel's seams: strips of 248 input columns (62 lanes of 4 + two halo lanes; 256 until round 2), bands
    of 16 output rows, widths that are not a multiple of 4 (the last words start at w - 4 and the per-lane byte selectors pick
    the clamped columns), 1- and 2-lane strips, images narrower than a dword (bytewise variant) or smaller than the 5x5 window"""
    img = _noise(w, h, w * 131 + h)
    assert np.array_equal(gpu_vs.pyr_down(img), oracle.pyr_down(img))


@pytest.mark.parametrize("w,h,pad_in,pad_out", [(250, 37, 3, 5), (124, 20, 1, 0), (1283, 21, 13, 7), (3, 9, 2, 1)])
def test_pyr_down_and_keyframe_through_padded_rows(gpu_vs, oracle, w, h, pad_in, pad_out):
    """the operator entry points take row strides (imgproc.hpp's buffers may be crops of larger images): rows that start at odd
    addresses, outputs with a pitch of their own; the padding bytes must neither be read into the result nor written"""
    import ctypes
    rng = np.random.default_rng(w * 17 + h)
    big = rng.integers(0, 256, (h, w + pad_in), dtype=np.uint8)
    img = np.ascontiguousarray(big[:, :w])
    ow, oh = w // 2, h // 2
    out = np.full((oh, ow + pad_out), 0xA5, np.uint8)
    lib = gpu_vs.lib()
    assert lib.vs_pyr_down(big.ctypes.data_as(ctypes.c_void_p), w, h, w + pad_in, out.ctypes.data_as(ctypes.c_void_p), ow, oh,
                           ow + pad_out, gpu_vs.MEM_HOST, None) == 0
    assert np.array_equal(out[:, :ow], oracle.pyr_down(img))
    assert (out[:, ow:] == 0xA5).all()
    if w >= 8:
        ts = gpu_vs.tile_size(w, h)
        tx, ty = w // ts, h // ts
        lmx, lmy = np.empty((2, ty, tx), np.uint16), np.empty((2, ty, tx), np.uint16)
        jx, jy = np.empty((4, ty, tx), np.float32), np.empty((4, ty, tx), np.float32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.vs_keyframe_fused(P(big), w, h, w + pad_in, ts, P(lmx), P(lmy), P(jx), P(jy), gpu_vs.MEM_HOST, None) == 0
        _, wlx, wly, wjx, wjy = gpu_vs.keyframe_fused(img)
        assert np.array_equal(lmx, wlx) and np.array_equal(lmy, wly) and np.array_equal(jx, wjx) and np.array_equal(jy, wjy)
        # bgr_image_warp and bgr_to_gray into a crop of a larger host image: the pixels right of every row stay untouched
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        t = gpu_vs.Transform.of(0.004, -0.003, 1.25, -0.75)
        for mode in (gpu_vs.WARP_LANCZOS2, gpu_vs.WARP_BILINEAR):
            dst = np.full((h, (w + pad_out) * 3), 0x5A, np.uint8)
            assert lib.vs_bgr_image_warp(P(bgr), w, h, 3 * w, 3, 8, ctypes.byref(t), mode, gpu_vs.BORDER_CLAMP, 255, P(dst), 3 * (w + pad_out),
                                         gpu_vs.MEM_HOST, None) == 0
            assert np.array_equal(dst[:, :3 * w].reshape(h, w, 3), gpu_vs.bgr_image_warp(bgr, t, mode=mode))
            assert (dst[:, 3 * w:] == 0x5A).all()
        gray = np.full((h, w + pad_out), 0x5A, np.uint8)
        assert lib.vs_bgr_to_gray(P(bgr), w, h, 3 * w, 8, 0, P(gray), w + pad_out, gpu_vs.MEM_HOST, None) == 0
        assert np.array_equal(gray[:, :w], gpu_vs.bgr_to_gray(bgr)) and (gray[:, w:] == 0x5A).all()


def test_pyr_down_known_answers(gpu_vs):
    # impulse responses of the integer form (sum w_i w_j in) >> 8
    img = np.zeros((32, 32), np.uint8)
    img[16, 16] = 255
    out = gpu_vs.pyr_down(img)
    assert out[8, 8] == 35 and out[8, 7] == 5 and out[8, 9] == 5 and out[7, 8] == 5 and out[9, 8] == 5
    assert out[7, 7] == 0
    img = np.zeros((32, 32), np.uint8)
    img[0, 0] = 255
    assert gpu_vs.pyr_down(img)[0, 0] == 120
    assert np.all(gpu_vs.pyr_down(np.full((40, 48), 77, np.uint8)) == 77)


@pytest.mark.parametrize("w,h", SIZES + [(5, 3)])
def test_grad_xy_bit_exact(gpu_vs, oracle, w, h):
    img = _noise(w, h, 2)
    gx, gy = gpu_vs.grad_xy(img)
    ox, oy = oracle.grad_xy(img)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy)


@pytest.mark.parametrize("w,h", SIZES)
def test_grad_argmax_bit_exact(gpu_vs, oracle, w, h):
    img = _img(w, h, 3)
    gx, gy = oracle.grad_xy(img)
    ts, lmx, lmy = gpu_vs.grad_argmax(gx, gy)
    ots, olx, oly = oracle.grad_argmax(gx, gy)
    assert ts == ots
    assert np.array_equal(lmx, olx) and np.array_equal(lmy, oly)


@pytest.mark.parametrize("ts", [2, 4, 7, 20, 33])
def test_grad_argmax_ties_and_zero_tiles(gpu_vs, oracle, ts):
    # flat image => all-zero gradients => top-left of every tile; plus planted equal maxima
    w, h = 10 * ts + 3, 6 * ts + 1
    gx = np.zeros((h, w), np.float32)
    gy = np.zeros((h, w), np.float32)
    gx[ts + 1, 2 * ts + 1] = 5.0
    gx[ts + 1, 2 * ts] = -5.0          # equal magnitude, smaller x on the same row wins
    if ts > 2:
        gy[3 * ts + 2, ts] = 7.0
        gy[3 * ts + 1, ts + 1] = 7.0   # smaller y wins over smaller x
    _, lmx, lmy = gpu_vs.grad_argmax(gx, gy, ts)
    _, olx, oly = oracle.grad_argmax(gx, gy, ts)
    assert np.array_equal(lmx, olx) and np.array_equal(lmy, oly)
    assert lmx[0, 0, 0] == 0 and lmx[1, 0, 0] == 0
    assert (lmx[0, 1, 2], lmx[1, 1, 2]) == (2 * ts, ts + 1)


@pytest.mark.parametrize("w,h", SIZES)
def test_sparse_jac_bit_exact(gpu_vs, oracle, w, h):
    img = _img(w, h, 4)
    gx, gy = oracle.grad_xy(img)
    _, lmx, lmy = oracle.grad_argmax(gx, gy)
    jx, jy = gpu_vs.sparse_jac(gx, gy, lmx, lmy)
    ojx, ojy = oracle.sparse_jac(gx, gy, lmx, lmy)
    assert np.array_equal(jx, ojx) and np.array_equal(jy, ojy)


@pytest.mark.parametrize("w,h", SIZES + [(120, 67), (60, 33)])
def test_keyframe_fused_equals_three_stage_oracle(gpu_vs, oracle, w, h):
    img = _img(w, h, 5)
    ts, lmx, lmy, jx, jy = gpu_vs.keyframe_fused(img)
    gx, gy = oracle.grad_xy(img)
    ots, olx, oly = oracle.grad_argmax(gx, gy)
    ojx, ojy = oracle.sparse_jac(gx, gy, olx, oly)
    assert ts == ots
    assert np.array_equal(lmx, olx) and np.array_equal(lmy, oly)
    assert np.array_equal(jx, ojx) and np.array_equal(jy, ojy)


@pytest.mark.parametrize("ts", [2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 3, 7, 21, 24, 64])
@pytest.mark.parametrize("w,h", [(324, 250), (257, 131), (1283, 97), (240, 64), (243, 61), (249, 40), (483, 23), (124, 21), (127, 22)])
def test_keyframe_fused_every_tile_size(gpu_vs, oracle, ts, w, h):
    """the ten strip kernels (the reference's tile sizes, CMakeLists.txt:212-253) and the generic kernel (any other size),
    on widths that leave partial strips, unaligned rows and remainder columns, and on widths at the strips' seams (a wave carries
    62 / (ts / 4 or 2) tiles: 240 or 248 columns for most sizes, 124 for ts = 2); tie-heavy content (values 0..7)"""
    if ts > min(w, h):
        pytest.skip("tile larger than the image")
    rng = np.random.default_rng(ts * 1000 + w)
    img = (rng.integers(0, 8, (h, w)) * 36).astype(np.uint8)
    _, lmx, lmy, jx, jy = gpu_vs.keyframe_fused(img, ts=ts)
    gx, gy = oracle.grad_xy(img)
    _, olx, oly = oracle.grad_argmax(gx, gy, ts=ts)
    ojx, ojy = oracle.sparse_jac(gx, gy, olx, oly)
    assert np.array_equal(lmx, olx) and np.array_equal(lmy, oly)
    assert np.array_equal(jx, ojx) and np.array_equal(jy, ojy)


def test_keyframe_fused_flat_and_saturated(gpu_vs, oracle):
    for img in (np.zeros((60, 80), np.uint8), np.full((60, 80), 255, np.uint8),
                (np.indices((60, 80)).sum(0) % 2 * 255).astype(np.uint8)):
        ts, lmx, lmy, jx, jy = gpu_vs.keyframe_fused(img)
        gx, gy = oracle.grad_xy(img)
        _, olx, oly = oracle.grad_argmax(gx, gy)
        ojx, ojy = oracle.sparse_jac(gx, gy, olx, oly)
        assert np.array_equal(lmx, olx) and np.array_equal(lmy, oly)
        assert np.array_equal(jx, ojx) and np.array_equal(jy, ojy)


TRANSFORMS = [(0, 0, 0, 0), (0.0, 0.0, 3.25, -2.5), (0.01, 0.005, -1.5, 2.75), (-0.02, 0.015, 9.0, -7.0),
              (0.0, 0.0, 1.0, 1.0), (0.05, -0.04, 40.0, 30.0)]


@pytest.mark.parametrize("w,h", [(640, 480), (322, 241), (80, 60)])
@pytest.mark.parametrize("tr", TRANSFORMS)
def test_sparse_warpdiff_bit_exact(gpu_vs, oracle, w, h, tr):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(w, h, 2, seed=6, path=[(0, 0, 0, 0), (0.002, -0.001, 1.3, -0.8)], margin=16)
    gx, gy = oracle.grad_xy(frames[1])
    _, lmx, lmy = oracle.grad_argmax(gx, gy)
    for lm in (lmx, lmy):
        g = gpu_vs.sparse_warpdiff(frames[0], frames[1], lm, gpu_vs.Transform.of(*tr))
        o = oracle.sparse_warpdiff(frames[0], frames[1], lm, oracle.Transform.of(*tr))
        assert np.array_equal(g, o)


def test_sparse_warpdiff_identity_is_zero(gpu_vs, oracle):
    # template == keyframe, identity T => every |diff| truncates to 0
    img = _img(320, 240, 8)
    gx, gy = oracle.grad_xy(img)
    _, lmx, _ = oracle.grad_argmax(gx, gy)
    assert not gpu_vs.sparse_warpdiff(img, img, lmx, gpu_vs.Transform.of()).any()


@pytest.mark.parametrize("w,h", [(640, 480), (160, 120)])
@pytest.mark.parametrize("tr", TRANSFORMS[:4])
def test_sparse_ica_matches_oracle(gpu_vs, oracle, w, h, tr):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(w, h, 2, seed=9, path=[(0, 0, 0, 0), (0.001, 0.002, -2.2, 1.1)], margin=16)
    gx, gy = oracle.grad_xy(frames[1])
    _, lmx, lmy = oracle.grad_argmax(gx, gy)
    jx, jy = oracle.sparse_jac(gx, gy, lmx, lmy)
    rng = np.random.default_rng(0)
    nt = lmx.shape[1] * lmx.shape[2]
    keep_x = rng.permutation(nt)[: int(nt * 0.8)]
    keep_y = rng.permutation(nt)[: int(nt * 0.8)]
    selx = lmx.reshape(2, -1)[:, keep_x]
    sely = lmy.reshape(2, -1)[:, keep_y]
    jacx = jx.reshape(4, -1)[:, keep_x]
    jacy = jy.reshape(4, -1)[:, keep_y]
    g = gpu_vs.sparse_ica(frames[0], frames[1], selx, sely, jacx, jacy, gpu_vs.Transform.of(*tr))
    o = oracle.sparse_ica(frames[0], frames[1], selx, sely, jacx, jacy, oracle.Transform.of(*tr))
    scale = np.abs(o).max() + 1e-30
    assert np.abs(g - o).max() <= 1e-12 * scale + 1e-9


@pytest.mark.parametrize("tr", TRANSFORMS)
def test_image_warp_bit_exact(gpu_vs, oracle, tr):
    img = _img(322, 241, 10)
    g = gpu_vs.image_warp(img, gpu_vs.Transform.of(*tr))
    o = oracle.image_warp(img, oracle.Transform.of(*tr))
    assert np.array_equal(g, o)


def test_image_warp_integer_shift_known_answer(gpu_vs):
    # align_test.cpp:358-400 / SURVEY 8c-7: 64x64, white 10x10 square at (20,20), T = (0,0,5,7);
    # ImageWarp(in, T.inverse()) moves the square by exactly (+5,+7)
    img = np.zeros((64, 64), np.uint8)
    img[20:30, 20:30] = 255
    t = gpu_vs.t_inverse(gpu_vs.Transform.of(0, 0, 5, 7))
    out = gpu_vs.image_warp(img, t)
    exp = np.zeros((64, 64), np.float32)
    exp[27:37, 25:35] = 255
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("tr", TRANSFORMS)
def test_bgr_image_warp_u8_bit_exact(gpu_vs, oracle, mode, border, tr):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(322, 241, 1, seed=11, channels=3, path=[(0, 0, 0, 0)], margin=8)
    g = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr), mode, border)
    o = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr), mode, border)
    assert np.array_equal(g, o)


@pytest.mark.parametrize("mode", [0, 1])
def test_bgr_image_warp_f32_and_u16(gpu_vs, oracle, mode):
    from video_stabilizer_amd import synth
    tr = (0.01, -0.004, 2.6, -1.7)
    f8, _ = synth.make_clip(200, 120, 1, seed=12, channels=3, path=[(0, 0, 0, 0)], margin=8)
    g = gpu_vs.bgr_image_warp(f8[0], gpu_vs.Transform.of(*tr), mode, 0, f32=True)
    o = oracle.bgr_image_warp(f8[0], oracle.Transform.of(*tr), mode, 0, f32=True)
    assert np.array_equal(g, o)
    f16, _ = synth.make_clip(200, 120, 1, seed=13, channels=3, bits=10, path=[(0, 0, 0, 0)], margin=8)
    assert f16.dtype == np.uint16 and f16.max() > 255
    g = gpu_vs.bgr_image_warp(f16[0], gpu_vs.Transform.of(*tr), mode, 1, max_value=1023)
    o = oracle.bgr_image_warp(f16[0], oracle.Transform.of(*tr), mode, 1, max_value=1023)
    assert np.array_equal(g, o)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("border", [0, 1])
@pytest.mark.parametrize("w,h", [(322, 241), (129, 50), (64, 16)])
def test_bgr_image_warp_u16_bit_exact(gpu_vs, oracle, mode, border, w, h):
    # 10-bit BGR in a u16 container (build extension): tuned path, odd sizes hit the tail stores
    rng = np.random.default_rng(w * 7 + h)
    src = rng.integers(0, 1024, (h, w, 3)).astype(np.uint16)
    for tr in TRANSFORMS:
        g = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode, border, max_value=1023)
        o = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), mode, border, max_value=1023)
        assert np.array_equal(g, o), tr
    full = rng.integers(0, 65536, (h, w, 3)).astype(np.uint16)
    g = gpu_vs.bgr_image_warp(full, gpu_vs.Transform.of(0.004, -0.003, 1.3, 2.6), mode, border)
    o = oracle.bgr_image_warp(full, oracle.Transform.of(0.004, -0.003, 1.3, 2.6), mode, border)
    assert np.array_equal(g, o)


@pytest.mark.parametrize("w,h", [(322, 241), (67, 35), (5, 9)])
def test_bgr_image_warp_u8_odd_sizes_and_large_rotation(gpu_vs, oracle, w, h):
    # tail stores, tiles narrower than 64, and footprints that do not fit the LDS window (global path)
    rng = np.random.default_rng(w + h)
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    for tr in [(0.0, 0.0, 0.5, 0.5), (0.3, 0.25, 3.0, -2.0), (-0.5, 0.0, 10.0, 5.0), (0.0, 1.0, 0.0, 0.0), (1.5, 0.0, 0.0, 0.0)]:
        for mode in (0, 1):
            for border in (0, 1):
                g = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode, border)
                o = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), mode, border)
                assert np.array_equal(g, o), (tr, mode, border)


def test_bgr_image_warp_gray_and_4ch(gpu_vs, oracle):
    rng = np.random.default_rng(5)
    tr = (0.003, 0.002, -3.3, 4.4)
    for c in (1, 2, 4):
        src = rng.integers(0, 256, (50, 70, c), dtype=np.uint8)
        g = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr))
        o = oracle.bgr_image_warp(src, oracle.Transform.of(*tr))
        assert np.array_equal(g, o)


def test_bgr_image_warp_batch(gpu_vs, oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(160, 96, 5, seed=14, channels=3, margin=16)
    ts = [(0.001 * i, -0.002 * i, 1.5 * i, -0.7 * i) for i in range(5)]
    g = gpu_vs.bgr_image_warp_batch(frames, [gpu_vs.Transform.of(*t) for t in ts])
    for i in range(5):
        o = oracle.bgr_image_warp(frames[i], oracle.Transform.of(*ts[i]))
        assert np.array_equal(g[i], o)


def test_bgr_image_warp_identity_is_round_trip(gpu_vs):
    # size-independent property at the BASELINE 4K size: identity transform, bilinear => output == input;
    # Lanczos2 at frac 0 has taps {-3.1e-5, 0.999861, -3.1e-5}: within 1 LSB of the input
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
    out = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(), gpu_vs.WARP_BILINEAR)
    assert np.array_equal(out, src)
    out = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(), gpu_vs.WARP_LANCZOS2)
    assert np.abs(out.astype(np.int16) - src.astype(np.int16)).max() <= 1


def test_bgr_image_warp_integer_shift_4k(gpu_vs):
    # integer translation, clamp border: bilinear output is the input shifted, exactly
    rng = np.random.default_rng(8)
    src = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
    out = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(0, 0, 7, -3), gpu_vs.WARP_BILINEAR)
    assert np.array_equal(out[3:, :-7], src[:-3, 7:])


@pytest.mark.parametrize("bits", [8, 10])
def test_bgr_to_gray_bit_exact(gpu_vs, oracle, bits):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(322, 241, 1, seed=15, channels=3, bits=bits, path=[(0, 0, 0, 0)], margin=8)
    assert np.array_equal(gpu_vs.bgr_to_gray(frames[0]), oracle.bgr_to_gray(frames[0]))


def test_bad_arguments_are_errors(gpu_vs):
    with pytest.raises(gpu_vs.VsError):
        gpu_vs.bgr_image_warp(np.zeros((4, 4, 5), np.uint8), gpu_vs.Transform.of())


@pytest.mark.parametrize("dtype,c", [(np.uint8, 3), (np.uint16, 3), (np.uint8, 1), (np.uint8, 4)])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("border", [0, 1])
def test_bgr_image_warp_window_equals_cropped_full_warp(gpu_vs, dtype, c, mode, border):
    """vs_bgr_image_warp_roi_batch (the stabilizer's crop, stabilizer.cpp:102-109): any window of the output, computed
    alone, is bit-identical to the same rows / columns of the full warp"""
    rng = np.random.default_rng(11)
    n, h, w = 3, 157, 211
    hi = 255 if dtype == np.uint8 else 1023
    src = rng.integers(0, hi + 1, (n, h, w, c)).astype(dtype)
    ts = [gpu_vs.Transform.of(0.01, -0.02, 3.3, -2.7), gpu_vs.Transform.of(-0.03, 0.015, -40.0, 25.5),
          gpu_vs.Transform.of(0.0, 0.0, 0.0, 0.0)]
    full = gpu_vs.bgr_image_warp_batch(src, ts, mode=mode, border=border, max_value=hi)
    for roi in [(0, 0, w, h), (32, 32, w - 64, h - 64), (1, 2, 64, 16), (5, 7, 65, 17), (w - 3, h - 2, 3, 2), (13, 0, 1, h),
                (0, 150, w, 1), (64, 16, 128, 32)]:
        x, y, rw, rh = roi
        got = gpu_vs.bgr_image_warp_roi_batch(src, ts, roi, mode=mode, border=border, max_value=hi)
        assert got.shape == (n, rh, rw, c)
        assert np.array_equal(got, full[:, y:y + rh, x:x + rw]), roi
    for bad in [(-1, 0, 4, 4), (0, 0, w + 1, 4), (w - 2, 0, 3, 4), (0, 0, 0, 4)]:
        with pytest.raises(gpu_vs.VsError):
            gpu_vs.bgr_image_warp_roi_batch(src, ts, bad)



def test_absurd_extents_and_inconsistent_arguments_are_refused_before_any_work(gpu_vs):
    """every call below would read or write far outside its (tiny) buffers if it were launched: extents whose products overflow int (w = 2^30 with 4
    channels: w * channels == 0), negative or zero sizes, strides shorter than a row, windows outside the frame, unknown modes / depths, null pointers.
    Each must come back VS_ERR_ARG (-1) from the host-side checks."""
    import ctypes as C
    L = gpu_vs.lib()
    buf = np.zeros(4096, np.uint8)
    p = C.c_void_p(buf.ctypes.data)
    t = gpu_vs.Transform.of()
    big = 1 << 30
    warp = lambda w, h, ss, c, bits, mode, border, ds, src=p, dst=p: L.vs_bgr_image_warp(src, w, h, ss, c, bits, C.byref(t), mode, border, 255, dst, ds, gpu_vs.MEM_HOST, None)
    bad = [
        warp(big, 4, 0, 4, 8, 0, 0, 0), warp(big, big, big, 1, 8, 0, 0, big), warp(70000, 2, 70000, 1, 8, 0, 0, 70000),
        warp(0, 4, 12, 3, 8, 0, 0, 12), warp(4, -1, 12, 3, 8, 0, 0, 12), warp(4, 4, 11, 3, 8, 0, 0, 12), warp(4, 4, 12, 3, 8, 0, 0, 11),
        warp(4, 4, 12, 3, 12, 0, 0, 12), warp(4, 4, 12, 3, 8, 99, 0, 12), warp(4, 4, 12, 3, 8, -1, 0, 12), warp(4, 4, 12, 3, 8, 0, 2, 12), warp(4, 4, 20, 5, 8, 0, 0, 20),
        warp(4, 4, 12, 3, 8, 0, 0, 12, src=None), warp(4, 4, 12, 3, 8, 0, 0, 12, dst=None),
        L.vs_bgr_image_warp_roi_batch(p, 48, 1, 4, 4, 12, 3, 8, C.byref(t), 0, 0, 255, 2, 2, 3, 3, p, 27, 9, gpu_vs.MEM_HOST, None),      # window leaves the frame
        L.vs_bgr_image_warp_roi_batch(p, 48, 1, 4, 4, 12, 3, 8, C.byref(t), 0, 0, 255, 0x7fffffff, 0, 2, 2, p, 12, 6, gpu_vs.MEM_HOST, None),
        L.vs_bgr_image_warp_batch(p, 10, 2, 4, 4, 12, 3, 8, C.byref(t), 0, 0, 255, p, 48, 12, gpu_vs.MEM_HOST, None),                        # frames overlap
        L.vs_pyr_down(p, big, big, big, p, big // 2, big // 2, big // 2, gpu_vs.MEM_HOST, None),
        L.vs_pyr_down(p, 8, 8, 4, p, 4, 4, 4, gpu_vs.MEM_HOST, None),
        L.vs_bgr_to_gray(p, big, 2, 3 * 4, 8, 0, p, 4, gpu_vs.MEM_HOST, None),
    ]
    assert bad == [-1] * len(bad), bad
    al = gpu_vs.Aligner(device=0)
    out, st = gpu_vs.Transform(), C.c_int32()
    batch = lambda n, w, h, stride, fmt, fs=0: L.vs_aligner_align_batch(al.h, p, fs, n, w, h, stride, fmt, gpu_vs.MEM_HOST, C.byref(al.params), C.byref(out), C.byref(st))
    bad = [batch(1, big, big, 0, gpu_vs.FMT_BGR8), batch(1, 70000, 16, 70000, gpu_vs.FMT_GRAY8), batch(1, 7, 64, 7, gpu_vs.FMT_GRAY8), batch(0, 64, 64, 64, gpu_vs.FMT_GRAY8),
           batch(1, 64, 64, 63, gpu_vs.FMT_GRAY8), batch(1, 64, 64, 64, 99), batch(2, 64, 64, 64, gpu_vs.FMT_GRAY8, fs=100)]
    assert bad == [-1] * len(bad), bad
    s = gpu_vs.Stabilizer(device=0)
    has, ow, oh = C.c_int32(), C.c_int(), C.c_int()
    proc = lambda n, w, h, stride, fmt: L.vs_stabilizer_process_batch(s.h, p, h * stride, n, w, h, stride, fmt, gpu_vs.MEM_HOST, p, 0, C.byref(has), C.byref(ow), C.byref(oh))
    bad = [proc(1, big, big, 0, gpu_vs.FMT_BGR8), proc(1, 70000, 70000, 3 * 70000, gpu_vs.FMT_BGR8), proc(1, 64, 64, 64, gpu_vs.FMT_GRAY8), proc(1, 64, 64, 100, gpu_vs.FMT_BGR8)]
    assert bad == [-1] * len(bad), bad
