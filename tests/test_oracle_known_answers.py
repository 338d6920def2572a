"""Pins the CPU oracle against hand-derivable answers taken from the reference's SOURCE TEXT
(SURVEY.md 8c items 1-10), not against itself.  The reference's own test (align_test.cpp) has no
numeric golden vectors for the Halide pipelines, and the reference cannot be built here, so these
known answers plus the property tests of align_test.cpp:261-601 are what pins the oracle.
"""
import math

import numpy as np
import pytest


# ---- 1. pyr_down = (sum_j sum_i w_j w_i in) >> 8, truncation, clamp-to-edge (generators.cpp:66-91)
def test_pyr_down_constant(oracle):
    for v in (0, 1, 77, 254, 255):
        assert np.all(oracle.pyr_down(np.full((20, 24), v, np.uint8)) == v)


def test_pyr_down_impulse_even(oracle):
    img = np.zeros((32, 32), np.uint8)
    img[16, 16] = 255                      # input (2X,2Y) = (16,16) -> output (8,8)
    out = oracle.pyr_down(img)
    assert out[8, 8] == (255 * 36) >> 8 == 35
    for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        assert out[8 + dy, 8 + dx] == (255 * 6) >> 8 == 5
    for dy, dx in ((1, 1), (1, -1), (-1, 1), (-1, -1)):
        assert out[8 + dy, 8 + dx] == (255 * 1) >> 8 == 0
    assert out.sum() == 35 + 4 * 5


def test_pyr_down_impulse_odd_x(oracle):
    img = np.zeros((32, 32), np.uint8)
    img[16, 17] = 255                      # odd x = 2X+1: outputs X and X+1 both see weight 4*6
    out = oracle.pyr_down(img)
    assert out[8, 8] == out[8, 9] == (255 * 24) >> 8 == 23


def test_pyr_down_clamp_to_edge_corner(oracle):
    img = np.zeros((32, 32), np.uint8)
    img[0, 0] = 255                        # counted with (1+4+6)^2 = 121
    assert oracle.pyr_down(img)[0, 0] == (255 * 121) >> 8 == 120


def test_pyr_down_matches_integer_formula_on_noise(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)
    pad = np.pad(img.astype(np.int64), 2, mode="edge")
    oh, ow = h // 2, w // 2
    exp = np.zeros((oh, ow), np.int64)
    for j in range(5):
        for i in range(5):
            exp += k[j] * k[i] * pad[j:j + 2 * oh:2, i:i + 2 * ow:2]
    assert np.array_equal(oracle.pyr_down(img), (exp >> 8).astype(np.uint8))


def test_pyr_down_output_extent_is_floor(oracle):
    assert oracle.pyr_down(np.zeros((67, 121), np.uint8)).shape == (33, 60)


# ---- 2. grad_xy (generators.cpp:215-223)
def test_grad_xy_ramp(oracle):
    img = np.tile(np.arange(40, dtype=np.uint8), (10, 1))
    gx, gy = oracle.grad_xy(img)
    assert np.all(gx[:, 1:-1] == 1.0) and np.all(gx[:, 0] == 0.5) and np.all(gx[:, -1] == 0.5)
    assert np.all(gy == 0.0)
    gx, gy = oracle.grad_xy(img.T.copy())
    assert np.all(gy[1:-1, :] == 1.0) and np.all(gy[0, :] == 0.5) and np.all(gx == 0.0)


def test_grad_xy_values_are_exact_halves(oracle):
    rng = np.random.default_rng(1)
    gx, gy = oracle.grad_xy(rng.integers(0, 256, (33, 47), dtype=np.uint8))
    assert np.all(gx * 2 == np.round(gx * 2)) and np.abs(gx).max() <= 127.5


# ---- 3. grad_argmax (generators.cpp:275-293) + tile-size rule (imgproc.cpp:151-162)
@pytest.mark.parametrize("w,h,ts", [(640, 480, 16), (320, 240, 8), (160, 120, 4), (80, 60, 2), (1280, 720, 20),
                                    (640, 360, 14), (1920, 1080, 20), (480, 270, 10), (3840, 2160, 20),
                                    (40, 30, 2), (240, 135, 4), (120, 67, 2), (60, 33, 2)])
def test_tile_size_rule(oracle, w, h, ts):
    assert oracle.tile_size(w, h) == ts


def test_grad_argmax_zero_tile_is_top_left(oracle):
    g = np.zeros((24, 36), np.float32)
    ts, lmx, lmy = oracle.grad_argmax(g, g, 4)
    xs, ys = np.meshgrid(np.arange(9) * 4, np.arange(6) * 4)
    assert np.array_equal(lmx[0], xs) and np.array_equal(lmx[1], ys)
    assert np.array_equal(lmy[0], xs) and np.array_equal(lmy[1], ys)


def test_grad_argmax_ties_smallest_y_then_x(oracle):
    g = np.zeros((16, 16), np.float32)
    g[5, 6] = -3.0
    g[5, 4] = 3.0       # same row, smaller x wins
    g[6, 1] = 3.0       # later row loses even with smaller x
    _, lmx, _ = oracle.grad_argmax(g, np.zeros_like(g), 8)
    assert (lmx[0, 0, 0], lmx[1, 0, 0]) == (4, 5)
    g2 = np.zeros((16, 16), np.float32)
    g2[3, 7] = 2.0
    g2[2, 7] = -2.0     # smaller y wins
    _, lmx, _ = oracle.grad_argmax(g2, np.zeros_like(g2), 8)
    assert (lmx[0, 0, 0], lmx[1, 0, 0]) == (7, 2)


def test_grad_argmax_remainder_pixels_ignored(oracle):
    # 640x360 with ts 14 -> 45x25 tiles cover 630x350: the last 10 columns / rows are never examined
    g = np.zeros((360, 640), np.float32)
    g[355, 635] = 100.0
    g[10, 635] = 100.0
    g[355, 10] = 100.0
    ts, lmx, _ = oracle.grad_argmax(g, g, 14)
    assert lmx.shape == (2, 25, 45)
    xs, ys = np.meshgrid(np.arange(45) * 14, np.arange(25) * 14)
    assert np.array_equal(lmx[0], xs) and np.array_equal(lmx[1], ys)


# ---- 4. lanczos2 polynomial (generators.cpp:31-47)
def test_lanczos2_values(oracle):
    assert oracle.lanczos2(0.0) == np.float32(0.999861)
    for x in (2.0, -2.0, 2.5, -3.0, 100.0):
        assert oracle.lanczos2(x) == 0.0
    assert oracle.lanczos2(1.999) != 0.0
    # weights at frac 0 / 0.25 / 0.5
    w0 = [oracle.lanczos2(u - 2 - 0.0) for u in range(5)]
    assert w0[0] == 0.0 and w0[4] == 0.0 and abs(w0[1] + 3.1e-5) < 2e-6 and abs(w0[3] + 3.1e-5) < 2e-6
    w25 = [oracle.lanczos2(u - 2 - 0.25) for u in range(5)]
    assert np.allclose(w25, [0, -0.084589, 0.877393, 0.235231, -0.017743], atol=2e-6)
    w5 = [oracle.lanczos2(u - 2 - 0.5) for u in range(5)]
    assert np.allclose(w5, [0, -0.063843, 0.573280, 0.573280, -0.063843], atol=2e-6)


def test_lanczos2_close_to_sinc_product(oracle):
    # lanczos2_opt.cpp:379-380: max |poly - sinc(x) sinc(x/2)| = 3.836e-4 on [-2,2]
    xs = np.linspace(-2, 2, 4001)
    ref = np.sinc(xs) * np.sinc(xs / 2)
    got = np.array([oracle.lanczos2(float(x)) for x in xs])
    err = np.abs(got - ref)[np.abs(xs) < 2].max()
    assert err <= 3.9e-4


def test_lanczos2_horner_order(oracle):
    # exact fp32 Horner in x*x with the printed coefficients, no FMA
    f = np.float32
    for x in (0.3, -1.25, 1.9):
        x = f(x)
        x2 = f(x * x)
        v = f(0.000858519)
        for c in (-0.0158853, 0.128693, -0.583468, 1.52229, -2.05238, 0.999861):
            v = f(f(c) + f(v * x2))
        assert oracle.lanczos2(float(x)) == v


# ---- 5. sparse kernels at identity (generators.cpp:646-700, 429-596)
def _keyframe(oracle, img):
    gx, gy = oracle.grad_xy(img)
    ts, lmx, lmy = oracle.grad_argmax(gx, gy)
    jx, jy = oracle.sparse_jac(gx, gy, lmx, lmy)
    return lmx, lmy, jx, jy


def test_warpdiff_identity_truncates_to_zero(oracle):
    from video_stabilizer_amd import synth
    img = synth.make_clip(160, 120, 1, seed=3, path=[(0, 0, 0, 0)], margin=8)[0][0]
    lmx, lmy, jx, jy = _keyframe(oracle, img)
    assert not oracle.sparse_warpdiff(img, img, lmx, oracle.Transform.of()).any()
    # the +-1 taps are -3.1e-5, not 0: ICA output is small but not exactly zero
    b = oracle.sparse_ica(img, img, lmx.reshape(2, -1), lmy.reshape(2, -1), jx.reshape(4, -1), jy.reshape(4, -1), oracle.Transform.of())
    bound = 0.03 * (np.abs(jx).sum() + np.abs(jy).sum())
    assert 0 < np.abs(b).max() < bound


def test_warpdiff_truncates_not_rounds(oracle):
    # generators.cpp:699 casts the float |diff| to u16: truncation.  A ramp of slope 4 sampled 0.4 px to the
    # right differs from the template by ~1.6 -> 1 (rounding would give 2); slope 4 at 0.2 px -> ~0.8 -> 0
    lm = np.stack([np.full((4, 4), 20, np.uint16), np.full((4, 4), 20, np.uint16)])
    ramp = np.tile((np.arange(40) * 4).astype(np.uint8), (40, 1))
    out = oracle.sparse_warpdiff_raw(ramp, ramp, lm, 0.0, 0.0, 0.4, 0.0)   # samples at x+0.4 -> +1.6 -> 1
    assert np.all(out == 1)
    assert not oracle.sparse_warpdiff_raw(ramp, ramp, lm, 0.0, 0.0, 0.2, 0.0).any()


def test_sparse_jac_formula(oracle):
    w, h = 64, 48
    gx = np.zeros((h, w), np.float32)
    gy = np.zeros((h, w), np.float32)
    gx[10, 20] = 3.5
    gy[30, 40] = -2.5
    lmx = np.array([[[20]], [[10]]], np.uint16)
    lmy = np.array([[[40]], [[30]]], np.uint16)
    jx, jy = oracle.sparse_jac(gx, gy, lmx, lmy)
    f = np.float32
    s = f(1.0) / f(w)
    assert jx[:, 0, 0].tolist() == [f(f(f(2) * f(3.5)) * f(20 - 32)) * s, f(f(f(2) * f(3.5)) * f(-(10 - 24))) * s, f(7.0), f(0)]
    assert jy[:, 0, 0].tolist() == [f(f(f(2) * f(-2.5)) * f(30 - 24)) * s, f(f(f(2) * f(-2.5)) * f(40 - 32)) * s, f(0), f(-5.0)]


# ---- 6. transform algebra: align_test.cpp:261-601, EPSILON 1e-5 (:249)
EPS = 1e-5


def _near(a, b):
    return abs(np.float32(a) - np.float32(b)) < EPS


def test_transform_inverse_cases(oracle):
    # align_test.cpp:265-279
    Ts = [(0, 0, 0, 0), (0.1, 0, 10, 20), (0, 0.1, 5, -5), (0.05, 0.05, 100, 50)]
    pts = [(0, 0), (100, 100), (50, 200), (-10, 30), (1.3, -2.7)]
    for T in Ts:
        t = oracle.Transform.of(*[np.float32(v) for v in T])
        ti = oracle.t_inverse(t)
        for p in pts:
            p = (float(np.float32(p[0])), float(np.float32(p[1])))
            wx, wy = oracle.t_warp(t, *p)
            ux, uy = oracle.t_warp(ti, wx, wy)
            assert _near(p[0], ux) and _near(p[1], uy)


def test_transform_compose_case(oracle):
    # align_test.cpp:313-345: T1.compose(T2)(p) == T2(T1(p))
    t1 = oracle.Transform.of(np.float32(0.1), 0, 10, 20)
    t2 = oracle.Transform.of(0, np.float32(0.1), 5, 5)
    t3 = oracle.t_compose(t1, t2)
    for p in [(0, 0), (10, 20), (50, 50), (-10, 30)]:
        a = oracle.t_warp(t3, *p)
        b = oracle.t_warp(t2, *oracle.t_warp(t1, *p))
        assert _near(a[0], b[0]) and _near(a[1], b[1])


def test_transform_randomized_properties(oracle):
    # align_test.cpp:444-601: the distributions' *properties* are the spec (sampled values are not portable)
    rng = np.random.default_rng(12345)

    def rt():
        return oracle.Transform.of(rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(-50, 50), rng.uniform(-50, 50))

    for _ in range(50):
        t = rt()
        ti = oracle.t_inverse(t)
        for _ in range(10):
            p = (rng.uniform(-200, 200), rng.uniform(-200, 200))
            u = oracle.t_warp(ti, *oracle.t_warp(t, *p))
            assert abs(u[0] - p[0]) < 1e-9 and abs(u[1] - p[1]) < 1e-9
    for _ in range(50):
        a, b, c = rt(), rt(), rt()
        ab_c = oracle.t_compose(oracle.t_compose(a, b), c)
        a_bc = oracle.t_compose(a, oracle.t_compose(b, c))
        assert np.allclose(ab_c.tup(), a_bc.tup(), atol=1e-9)
        ident = oracle.t_compose(a, oracle.t_inverse(a))
        assert np.allclose(ident.tup(), (0, 0, 0, 0), atol=1e-9)
        ident = oracle.t_compose(oracle.t_inverse(a), a)
        assert np.allclose(ident.tup(), (0, 0, 0, 0), atol=1e-9)


def test_warp_about_center_and_corner_displacement(oracle):
    t = oracle.Transform.of(0.0, 0.0, 3.0, -4.0)
    assert oracle.t_max_corner_displacement(t, 640, 480) == 5.0
    # pure rotation by B about (w/2,h/2): centre does not move (imgproc.cpp:401-411)
    r = oracle.Transform.of(0.0, 0.1, 0.0, 0.0)
    assert oracle.t_warp(r, 320, 240, center=(320, 240)) == (320.0, 240.0)
    # corners (0,0),(w,0),(0,h),(w,h) about (w/2,h/2): all move by |B| * half-diagonal
    assert abs(oracle.t_max_corner_displacement(r, 640, 480) - 0.1 * 400) < 1e-12
    assert oracle.t_warp(r, 1, 2, center=(0, 0)) == oracle.t_warp(r, 1, 2)


def test_ul_conversion_centres(oracle):
    # sparse kernels: centre (w/2,h/2) (imgproc.cpp:72-75); image_warp: ((w-1)/2,(h-1)/2) (imgproc.cpp:125-129)
    t = oracle.Transform.of(0.01, 0.02, 3.0, 4.0)
    p = oracle.ul_params_sparse(t, 640, 480)
    assert p[2] == np.float32(3.0 - 0.01 * 320 + 0.02 * 240) and p[3] == np.float32(4.0 - 0.02 * 320 - 0.01 * 240)
    q = oracle.ul_params_warp(t, 640, 480)
    assert q[2] == np.float32(3.0 - 0.01 * 319.5 + 0.02 * 239.5) and q[3] == np.float32(4.0 - 0.02 * 319.5 - 0.01 * 239.5)


# ---- 7. ImageWarp integer shift (align_test.cpp:358-400, stronger than its +-0.5 px check)
def test_image_warp_integer_shift(oracle):
    img = np.zeros((64, 64), np.uint8)
    img[20:30, 20:30] = 255
    out = oracle.image_warp(img, oracle.t_inverse(oracle.Transform.of(0, 0, 5, 7)))
    exp = np.zeros((64, 64), np.float32)
    exp[27:37, 25:35] = 255
    assert np.array_equal(out, exp)


def test_image_warp_half_pixel_is_average(oracle):
    img = np.tile((np.arange(32) * 8).astype(np.uint8), (8, 1))
    out = oracle.image_warp_raw(img, 0.0, 0.0, 0.5, 0.0)
    assert np.all(out[:, :-1] == img[:, :-1] + 4.0)
    assert np.all(out[:, -1] == img[:, -1])          # clamp-to-edge


# ---- bgr_image_warp (build-defined): consistency with the two reference samplers
def test_bgr_warp_bilinear_equals_image_warp_per_channel(oracle):
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    t = oracle.Transform.of(0.01, -0.02, 1.7, -2.2)
    f = oracle.bgr_image_warp(src, t, oracle.WARP_BILINEAR, oracle.BORDER_CLAMP, f32=True)
    for c in range(3):
        assert np.array_equal(f[:, :, c], oracle.image_warp(np.ascontiguousarray(src[:, :, c]), t))


def test_bgr_warp_lanczos_equals_sparse_sampler(oracle):
    # the full-frame Lanczos2 sample at pixel p equals what sparse_warpdiff sees there:
    # |sample - template| with template = 0 is trunc(sample)
    rng = np.random.default_rng(3)
    gray = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    A, B, TX, TY = 0.01, 0.005, 2.3, -1.4
    ys, xs = np.meshgrid(np.arange(4, 44, 8), np.arange(4, 60, 8), indexing="ij")
    lm = np.stack([xs, ys]).astype(np.uint16)
    wd = oracle.sparse_warpdiff_raw(np.zeros_like(gray), gray, lm, A, B, TX, TY)
    # build a centre-based transform whose UL conversion (about (w-1)/2) reproduces (A,B,TX,TY)
    cx, cy = (64 - 1) * 0.5, (48 - 1) * 0.5
    t = oracle.Transform.of(np.float32(A), np.float32(B), TX + np.float32(A) * cx - np.float32(B) * cy, TY + np.float32(B) * cx + np.float32(A) * cy)
    p = oracle.ul_params_warp(t, 64, 48)
    if (p[2], p[3]) == (np.float32(TX), np.float32(TY)):
        f = oracle.bgr_image_warp(gray[:, :, None], t, oracle.WARP_LANCZOS2, oracle.BORDER_CLAMP, f32=True)[:, :, 0]
        assert np.array_equal(np.floor(np.abs(f[ys, xs])).astype(np.uint16), wd)


def _round_f32(q):
    """a Fraction rounded to the nearest float32, ties to even -- exact integer arithmetic, no intermediate double"""
    from fractions import Fraction
    if q == 0:
        return np.float32(0.0)
    sign = -1 if q < 0 else 1
    q = abs(q)
    e = q.numerator.bit_length() - q.denominator.bit_length()          # 2^(e-1) < q < 2^(e+1)
    if Fraction(2) ** e > q:
        e -= 1                                                             # now 2^e <= q < 2^(e+1)
    e = max(e, -126)                                                       # subnormals share the exponent of the smallest normal
    scaled = q / Fraction(2) ** (e - 23)                                   # the significand as a rational in [2^23, 2^24)
    n, rem = divmod(scaled.numerator, scaled.denominator)
    twice = 2 * rem
    if twice > scaled.denominator or (twice == scaled.denominator and (n & 1)):
        n += 1
    return np.float32(sign * float(Fraction(n) * Fraction(2) ** (e - 23)))    # n * 2^(e-23) is a float32 value: exact in a double


def test_contracted_lanczos2_twin_equals_a_literal_restatement_with_exact_fma(oracle):
    # VSO_WARP_LANCZOS2_CONTRACTED = generators.cpp:31-47 + :684-697 with the multiply-adds fused where LLVM may fuse them on the
    # reference's target (CMakeLists.txt:151: "fma", no strict_float anywhere): c + val*x2 -> fma(val, x2, c); sum_num += w2d*val
    # -> fma(w2d, val, sum_num); w2d = wx*wy, sum_den += w2d and the divide as written.  Restated here with an fma that is
    # exact by construction (rationals, one rounding), independent of std::fmaf and of the machine.
    from fractions import Fraction as F
    f32 = np.float32

    def fma(a, b, c):
        return _round_f32(F(float(a)) * F(float(b)) + F(float(c)))

    def lz(x):
        x = f32(x)
        x2 = f32(x * x)
        v = f32(0.000858519)
        for c in (-0.0158853, 0.128693, -0.583468, 1.52229, -2.05238, 0.999861):
            v = fma(v, x2, f32(c))
        return f32(0.0) if abs(x) >= 2.0 else v

    rng = np.random.default_rng(11)
    h, w = 7, 9
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    src[2, 3] = (255, 0, 255)
    src[3, 3] = (0, 255, 0)
    for border in (oracle.BORDER_CLAMP, oracle.BORDER_CONSTANT):
        t = oracle.Transform.of(0.03, -0.02, 0.7, -1.3)
        got = oracle.bgr_image_warp(src, t, oracle.WARP_LANCZOS2_CONTRACTED, border, f32=True)
        p = oracle.ul_params_warp(t, w, h)
        A, B, TX, TY = (f32(v) for v in p)
        A1 = f32(f32(1.0) + A)
        for y in range(h):
            for x in range(w):
                Wx = f32(f32(f32(A1 * f32(x)) - f32(B * f32(y))) + TX)          # generators.cpp:141 (un-contracted: same positions as VSO_WARP_LANCZOS2)
                Wy = f32(f32(f32(B * f32(x)) + f32(A1 * f32(y))) + TY)
                flx, fly = f32(np.floor(Wx)), f32(np.floor(Wy))
                frx, fry = f32(Wx - flx), f32(Wy - fly)
                wx = [lz(f32(f32(u - 2) - frx)) for u in range(5)]
                wy = [lz(f32(f32(u - 2) - fry)) for u in range(5)]
                for c in range(3):
                    num, den = f32(0.0), f32(0.0)
                    for ry in range(5):
                        for rx in range(5):
                            sx, sy = int(flx) + rx - 2, int(fly) + ry - 2
                            if border == oracle.BORDER_CONSTANT and (sx < 0 or sy < 0 or sx >= w or sy >= h):
                                val = f32(0.0)
                            else:
                                val = f32(src[min(max(sy, 0), h - 1), min(max(sx, 0), w - 1), c])
                            w2d = f32(wx[rx] * wy[ry])
                            num = fma(w2d, val, num)
                            den = f32(den + w2d)
                    assert f32(num / den) == got[y, x, c], (border, y, x, c)
    # ... and it is a different function from the un-contracted mode (else the twin would test nothing), within one LSB of it
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    t = oracle.Transform.of(0.01, 0.004, 2.3, -1.6)
    e = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2, f32=True)
    k = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2_CONTRACTED, f32=True)
    assert not np.array_equal(e, k) and np.abs(e - k).max() < 1e-3
    ei = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2).astype(int)
    ki = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2_CONTRACTED).astype(int)
    assert np.abs(ei - ki).max() <= 1


def test_separable_lanczos2_twin_equals_a_literal_restatement_with_exact_fma(oracle):
    # VSO_WARP_LANCZOS2_SEPARABLE = the contracted weights (fma Horner, generators.cpp:38-46) and the 4 x 4 live taps of generators.cpp:684-697,
    # summed rows first, then columns: h[ry] = fma(wx4, v4, fma(wx3, v3, fma(wx2, v2, wx1 * v1))), num = fma(wy4, h4, ... wy1 * h1),
    # den = ((wx1 + wx2) + (wx3 + wx4)) * ((wy1 + wy2) + (wy3 + wy4)), out = num * RN(1 / den).  Restated here with an fma and a reciprocal that
    # are exact by construction (rationals, one rounding each), independent of std::fmaf, of the C compiler's divide and of the machine.
    from fractions import Fraction as F
    f32 = np.float32

    def fma(a, b, c):
        return _round_f32(F(float(a)) * F(float(b)) + F(float(c)))

    def lz(x):
        x = f32(x)
        x2 = f32(x * x)
        v = f32(0.000858519)
        for c in (-0.0158853, 0.128693, -0.583468, 1.52229, -2.05238, 0.999861):
            v = fma(v, x2, f32(c))
        return f32(0.0) if abs(x) >= 2.0 else v

    rng = np.random.default_rng(12)
    h, w = 7, 9
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    src[2, 3] = (255, 0, 255)
    src[3, 3] = (0, 255, 0)
    for border in (oracle.BORDER_CLAMP, oracle.BORDER_CONSTANT):
        for tr in ((0.03, -0.02, 0.7, -1.3), (0.0, 0.0, 1.0, -2.0)):          # (the second: every fraction exactly 0 -- tap 4's select at work)
            t = oracle.Transform.of(*tr)
            got = oracle.bgr_image_warp(src, t, oracle.WARP_LANCZOS2_SEPARABLE, border, f32=True)
            A, B, TX, TY = (f32(v) for v in oracle.ul_params_warp(t, w, h))
            A1 = f32(f32(1.0) + A)
            for y in range(h):
                for x in range(w):
                    Wx = f32(f32(f32(A1 * f32(x)) - f32(B * f32(y))) + TX)      # generators.cpp:141 (the same positions as every other mode)
                    Wy = f32(f32(f32(B * f32(x)) + f32(A1 * f32(y))) + TY)
                    flx, fly = f32(np.floor(Wx)), f32(np.floor(Wy))
                    frx, fry = f32(Wx - flx), f32(Wy - fly)
                    wx = [lz(f32(f32(u - 1) - frx)) for u in range(4)]           # taps 1..4 of the 5-tap window (tap 0 weighs exactly 0)
                    wy = [lz(f32(f32(u - 1) - fry)) for u in range(4)]
                    den = f32(f32(f32(wx[0] + wx[1]) + f32(wx[2] + wx[3])) * f32(f32(wy[0] + wy[1]) + f32(wy[2] + wy[3])))
                    rden = _round_f32(1 / F(float(den)))                           # the correctly rounded reciprocal
                    for c in range(3):
                        hrow = []
                        for ry in range(4):
                            v = []
                            for rx in range(4):
                                sx, sy = int(flx) + rx - 1, int(fly) + ry - 1
                                if border == oracle.BORDER_CONSTANT and (sx < 0 or sy < 0 or sx >= w or sy >= h):
                                    v.append(f32(0.0))
                                else:
                                    v.append(f32(src[min(max(sy, 0), h - 1), min(max(sx, 0), w - 1), c]))
                            hrow.append(fma(wx[3], v[3], fma(wx[2], v[2], fma(wx[1], v[1], f32(wx[0] * v[0])))))
                        num = fma(wy[3], hrow[3], fma(wy[2], hrow[2], fma(wy[1], hrow[1], f32(wy[0] * hrow[0]))))
                        assert f32(num * rden) == got[y, x, c], (border, tr, y, x, c)
    # ... a different function from the contracted and the un-contracted forms (else the twin would test nothing), within one LSB of both
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    t = oracle.Transform.of(0.01, 0.004, 2.3, -1.6)
    e = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2, f32=True)
    k = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2_CONTRACTED, f32=True)
    sp = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2_SEPARABLE, f32=True)
    assert not np.array_equal(sp, k) and not np.array_equal(sp, e) and np.abs(sp - e).max() < 1e-3
    ei = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2).astype(int)
    si = oracle.bgr_image_warp(big, t, oracle.WARP_LANCZOS2_SEPARABLE).astype(int)
    assert np.abs(ei - si).max() <= 1


def test_round_f32_helper():
    from fractions import Fraction as F
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a, b, c = (np.float32(v) for v in rng.normal(size=3) * 10.0 ** rng.integers(-3, 4))
        exact = F(float(a)) * F(float(b)) + F(float(c))
        r = _round_f32(exact)
        lo, hi = np.nextafter(r, np.float32(-np.inf)), np.nextafter(r, np.float32(np.inf))
        assert abs(F(float(r)) - exact) <= abs(F(float(lo)) - exact) and abs(F(float(r)) - exact) <= abs(F(float(hi)) - exact)
    assert _round_f32(F(1) + F(1, 2 ** 24)) == np.float32(1.0)                     # tie -> even
    assert _round_f32(F(1) + F(3, 2 ** 24)) == np.float32(1.0) + np.float32(2.0 ** -22)   # tie -> even (up)
    assert _round_f32(F(1, 2 ** 149)) == np.float32(2.0 ** -149) and _round_f32(F(1, 2 ** 151)) == np.float32(0.0)


def test_bgr_warp_store_rule_round_half_up_saturate(oracle):
    src = np.zeros((8, 8, 1), np.uint8)
    src[:, 4:] = 255
    out = oracle.bgr_image_warp(src, oracle.Transform.of(0, 0, 0.5, 0), oracle.WARP_BILINEAR)
    assert out[4, 3, 0] == 128          # 127.5 rounds up
    hi = np.full((8, 8, 1), 255, np.uint8)
    hi[:, 4] = 0
    out = oracle.bgr_image_warp(hi, oracle.Transform.of(0, 0, 0.5, 0), oracle.WARP_LANCZOS2)
    assert out.max() == 255             # Lanczos overshoot saturates


def test_bgr_warp_constant_border_is_black(oracle):
    src = np.full((16, 16, 3), 200, np.uint8)
    out = oracle.bgr_image_warp(src, oracle.Transform.of(0, 0, 8, 0), oracle.WARP_BILINEAR, oracle.BORDER_CONSTANT)
    assert np.all(out[:, :7] == 200) and np.all(out[:, 8:] == 0)
    out = oracle.bgr_image_warp(src, oracle.Transform.of(0, 0, 8, 0), oracle.WARP_BILINEAR, oracle.BORDER_CLAMP)
    assert np.all(out == 200)


def test_bgr_to_gray_weights(oracle):
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30]]], np.uint8)
    g = oracle.bgr_to_gray(px)[0]
    exp = [(255 * 3735 + 16384) >> 15, (255 * 19235 + 16384) >> 15, (255 * 9798 + 16384) >> 15, 255,
           (10 * 3735 + 20 * 19235 + 30 * 9798 + 16384) >> 15]
    assert g.tolist() == exp


# ---- selection / Hessian / solve (alignment.cpp:435-583)
def test_select_count_is_float_product(oracle):
    # selected_count = size_t(size * 0.8f): 1200 -> 960, 5184 -> 4147, 20736 -> 16588
    for (tx, ty), n in (((40, 30), 960), ((96, 54), 4147), ((192, 108), 16588), ((20, 15), 240), ((45, 44), 1584)):
        wd = np.random.default_rng(tx).integers(0, 30, (ty, tx)).astype(np.uint16)
        idx = oracle.select_smallest(wd, 0.8)
        assert len(idx) == n and len(set(idx.tolist())) == n
        thr = np.sort(wd.ravel())[n - 1]
        kept = wd.ravel()[idx]
        assert kept.max() <= thr and (wd.ravel() < thr).sum() <= n
        assert set(np.flatnonzero(wd.ravel() < thr).tolist()) <= set(idx.tolist())


def test_hessian_and_inverse(oracle):
    rng = np.random.default_rng(4)
    jx = rng.normal(size=(4, 300)).astype(np.float32)
    jy = rng.normal(size=(4, 280)).astype(np.float32)
    H = oracle.hessian(jx, jy)
    exp = jx.astype(np.float64) @ jx.astype(np.float64).T + jy.astype(np.float64) @ jy.astype(np.float64).T
    assert np.allclose(H, exp, rtol=1e-12) and np.array_equal(H, H.T)
    cond, H2, Hinv = oracle.condition_and_invert(H)
    sv = np.linalg.svd(exp, compute_uv=False)
    assert abs(cond - sv[0] / (sv[-1] + 1e-10)) < 1e-9 * cond
    assert np.array_equal(H2, H)        # well conditioned: no Tikhonov term
    assert np.allclose(Hinv, np.linalg.inv(exp), rtol=1e-9, atol=1e-12)


def test_ill_conditioned_gets_tikhonov(oracle):
    H = np.diag([1e8, 1.0, 1.0, 1e-3])
    cond, H2, Hinv = oracle.condition_and_invert(H)
    assert cond > 1e6
    assert np.allclose(np.diag(H2), np.diag(H) + 1e-6 * 1e8)
    assert np.allclose(Hinv, np.linalg.inv(H2), rtol=1e-9)


# ---- 9. tvl1_smooth (smoother.cpp:18-65) and L1SmootherCenter (:67-127)
def test_tvl1_constant_and_small_steps(oracle):
    assert np.all(oracle.tvl1_smooth(np.full(9, 3.25), 4.0) == 3.25)
    # |diff| <= lambda everywhere: every sweep clamps neighbours to midpoints
    x = oracle.tvl1_smooth(np.array([0.0, 1.0, 0.0, 1.0]), 4.0)
    assert x.std() < 0.5


def test_tvl1_matches_literal_python(oracle):
    def lit(data, lam, iters=100):
        x = list(data)
        n = len(x)
        for _ in range(iters):
            for i in range(n):
                x[i] = (1.0 - 0.5) * x[i] + 0.5 * data[i]
            for i in range(n - 1):
                d = x[i + 1] - x[i]
                m = abs(d)
                if m > lam:
                    s = (m - lam) / m * 0.5
                    x[i] += d * s
                    x[i + 1] -= d * s
                else:
                    mid = 0.5 * (x[i] + x[i + 1])
                    x[i] = mid
                    x[i + 1] = mid
        return x
    rng = np.random.default_rng(5)
    for lam in (0.5, 4.0):
        d = rng.normal(scale=6.0, size=16)
        assert oracle.tvl1_smooth(d, lam).tolist() == lit(d.tolist(), lam)


def test_smoother_window_and_lag(oracle):
    sm = oracle.Smoother(10, 5, 4.0)
    rng = np.random.default_rng(6)
    meas = [oracle.Transform.of(*rng.normal(size=4)) for _ in range(30)]
    outs = []
    for m in meas:
        outs.append(sm.update(m))
    assert [o[0] for o in outs[:5]] == [False] * 5 and all(o[0] for o in outs[5:])
    # update m finalises index k = m-5 over the window [max(0,k-10), k+5]
    for m in (5, 12, 29):
        k = m - 5
        lo = max(0, k - 10)
        tx = oracle.tvl1_smooth(np.array([t.TX for t in meas[lo:k + 6]]), 4.0)
        assert outs[m][1].TX == tx[k - lo]


# ---- 8 / 10. end to end on a known motion; damped step
def test_align_first_call_false_then_recovers_translation(oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 480, 2, seed=12345, path=[(0, 0, 0, 0), (0, 0, 3.25, -2.5)])
    al = oracle.Aligner()
    ok, t = al.align_next(frames[0])
    assert ok is False and t.tup() == (0, 0, 0, 0) and al.debug().fail_reason == 1
    ok, t = al.align_next(frames[1])
    assert ok and al.debug().levels == 5
    # frame1(p) = frame0(p + d) with d = (3.25,-2.5); the loop stops with <= ~0.25 px left
    assert math.hypot(t.TX + 3.25, t.TY - 2.5) < 0.3 and abs(t.A) < 2e-3 and abs(t.B) < 2e-3


def test_pyramid_levels_rule(oracle):
    # halve until < pyramid_min (alignment.cpp:164-169): defaults give 6 levels at 1080p, min_width 256 gives 3
    from video_stabilizer_amd import synth
    f, _ = synth.make_clip(960, 540, 1, seed=1, path=[(0, 0, 0, 0)], margin=4)
    al = oracle.Aligner()
    al.align_next(f[0])
    assert al.debug().levels == 5
    al = oracle.Aligner(pyramid_min_width=256)
    al.align_next(np.zeros((1080, 1920), np.uint8))
    assert al.debug().levels == 3


def test_first_gn_step_is_quarter_of_least_squares(oracle):
    # Jacobians carry a factor 2 and sparse_ica halves its output (generators.cpp:369-384,595), so
    # Hinv*b = 1/4 of the Gauss-Newton step computed independently from unscaled gradients.
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 2, seed=9, path=[(0, 0, 0, 0), (0, 0, 0.6, -0.4)], margin=16)
    tmpl, key = frames[0], frames[1]
    lmx, lmy, jx, jy = _keyframe(oracle, key)
    sx, sy = lmx.reshape(2, -1), lmy.reshape(2, -1)
    JX, JY = jx.reshape(4, -1), jy.reshape(4, -1)
    b = oracle.sparse_ica(tmpl, key, sx, sy, JX, JY, oracle.Transform.of())
    _, _, Hinv = oracle.condition_and_invert(oracle.hessian(JX, JY))
    step = Hinv @ b
    # independent least squares on the same points: residual r = tmpl(p) - key(p), columns g = J/2
    G = np.concatenate([JX, JY], 1).astype(np.float64).T / 2.0
    r = np.concatenate([tmpl[sx[1], sx[0]].astype(np.float64) - key[sx[1], sx[0]], tmpl[sy[1], sy[0]].astype(np.float64) - key[sy[1], sy[0]]])
    full = np.linalg.lstsq(G, r, rcond=None)[0]
    assert np.allclose(step, full / 4.0, rtol=2e-3, atol=5e-6)


def test_stabilizer_lag_and_static_clip(oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(160, 128, 1, seed=4, channels=3, path=[(0, 0, 0, 0)], margin=4)
    st = oracle.Stabilizer(crop_pixels=8, warp_mode=oracle.WARP_BILINEAR)
    outs = [st.process(frames[0]) for _ in range(14)]
    assert all(o is None for o in outs[:10]) and all(o is not None for o in outs[10:])   # empty for the first `lag`
    # static clip: every measurement is ~identity, correction is identity, output = cropped input
    m, a, ok = st.state()
    assert ok and max(abs(v) for v in m.tup()) < 1e-3   # not exactly 0: the frac-0 Lanczos taps are -3.1e-5, not 0
    assert np.array_equal(outs[-1], frames[0][8:-8, 8:-8])


def test_row_parallel_mode_changes_nothing(oracle):
    """oracle.set_threads(n): the image-sized stage loops run row-parallel (CPU-baseline timing mode); no
    reduction is split, so every output is bit-identical to the serial run"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(322, 246, 3, seed=19, channels=3)

    def run():
        a = oracle.Aligner()
        out = []
        for f in frames:
            ok, t = a.align_next(f)
            out.append((ok, t.tup(), oracle.bgr_image_warp(f, t if ok else oracle.Transform.of(0.01, -0.02, 1.5, 2.5))))
        lv = a.level(0)
        return out, lv["argmax"], lv["jac"]

    try:
        oracle.set_threads(1)
        r1, am1, j1 = run()
        oracle.set_threads(5)
        assert oracle.lib().vso_get_threads() == 5
        r5, am5, j5 = run()
    finally:
        oracle.set_threads(1)
    for a, b in zip(r1, r5):
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])
    for s in (0, 1):
        assert np.array_equal(am1[s], am5[s]) and np.array_equal(j1[s], j5[s])


def test_stable_selection_rule(oracle):
    """vso_select_smallest_stable: the documented STL-independent rule against a literal restatement, and
    against std::nth_element's outcome as a multiset of abs_delta values (any conforming nth_element keeps the same VALUES)"""
    rng = np.random.default_rng(8)
    for (ty, tx, hi, frac) in ((9, 16, 12, 0.8), (27, 48, 4, 0.8), (5, 7, 60000, 0.8), (12, 12, 300, 0.5), (3, 3, 2, 1.0), (4, 4, 9, 0.01)):
        wd = rng.integers(0, hi, size=(ty, tx)).astype(np.uint16)
        flat = wd.ravel()
        k = int(np.float32(flat.size) * np.float32(frac)) if frac < 1.0 else flat.size
        got = oracle.select_smallest_stable(wd, frac)
        want = sorted(sorted(range(flat.size), key=lambda i: (int(flat[i]), i))[:len(got)])
        assert list(got) == want and np.all(np.diff(got) > 0)
        stl = oracle.select_smallest(wd, frac)
        assert len(stl) == len(got) and sorted(flat[stl].tolist()) == sorted(flat[got].tolist())
