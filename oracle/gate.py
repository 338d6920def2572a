"""SURVEY 8(d)'s tolerance gate for bgr_image_warp outputs, as code (test infrastructure, like everything under oracle/: only
tests/, __graft_entry__.smoke() and bench.py's parity leg import it).

  integer modes:  max |d| <= 1 LSB  and  identical fraction >= 0.9999
  float mode:     |d| <= eps * sum|w v| / |sum w|   per output sample ("1 ULP-equivalent": eps = 2^-23, the spacing of fp32
                  numbers in [1, 2); the right-hand side is the magnitude the sampler's rounding errors scale with)

`lanczos_bound` evaluates the right-hand side for a frame and a transform: the fp32 sampling positions exactly as image_warp
computes them (generators.cpp:141-152 with imgproc.cpp:125-131's parameters), the polynomial weights (generators.cpp:31-47), the
4 x 4 live taps with clamp-to-edge (generators.cpp:672-697), the two sums in float64.
"""
import numpy as np

from . import oracle as O

EPS32 = 2.0 ** -23
INTEGER_IDENTICAL_MIN = 0.9999


def _lanczos2_f32(x):
    x = x.astype(np.float32)
    x2 = x * x
    v = np.full_like(x2, np.float32(0.000858519))
    for c in (-0.0158853, 0.128693, -0.583468, 1.52229, -2.05238, 0.999861):
        v = np.float32(c) + v * x2
    return np.where(np.abs(x) >= np.float32(2.0), np.float32(0.0), v)


def lanczos_bound(src, tr, rows=None, eps=EPS32):
    """eps * sum|w v| / |sum w| for every output sample of bgr_image_warp(src, tr) (clamp border), shape (rows, w, c) float64"""
    h, w, c = src.shape
    p = O.ul_params_warp(O.Transform.of(*tr), w, h)
    A1 = np.float32(1.0) + np.float32(p[0])
    B, TX, TY = np.float32(p[1]), np.float32(p[2]), np.float32(p[3])
    r0, r1 = rows if rows else (0, h)
    xs = np.arange(w, dtype=np.float32)[None, :]
    ys = np.arange(r0, r1, dtype=np.float32)[:, None]
    Wx = (A1 * xs - B * ys) + TX
    Wy = (B * xs + A1 * ys) + TY
    flx, fly = np.floor(Wx), np.floor(Wy)
    frx, fry = Wx - flx, Wy - fly
    ix, iy = flx.astype(np.int64), fly.astype(np.int64)
    num = np.zeros((r1 - r0, w, c), np.float64)
    den = np.zeros((r1 - r0, w), np.float64)
    s = src.astype(np.float64)
    wxs = [_lanczos2_f32(np.float32(u - 2) - frx) for u in range(1, 5)]
    wys = [_lanczos2_f32(np.float32(u - 2) - fry) for u in range(1, 5)]
    for ry in range(4):
        sy = np.clip(iy + ry - 1, 0, h - 1)
        for rx in range(4):
            sx = np.clip(ix + rx - 1, 0, w - 1)
            w2d = (wxs[rx] * wys[ry]).astype(np.float64)
            num += np.abs(w2d)[..., None] * s[sy, sx]
            den += w2d
    return eps * num / np.abs(den)[..., None]


def integer_gate(got, want):
    """(pass, {max_abs_diff, identical_fraction}) for two integer frames"""
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    mx, same = int(d.max()), float((d == 0).mean())
    return (mx <= 1 and same >= INTEGER_IDENTICAL_MIN), {"max_abs_diff_lsb": mx, "identical_fraction": same}


def float_gate(got, want, bound):
    """(pass, {worst |d| / bound, fraction within}) for two float frames and lanczos_bound of the same rows"""
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    ratio = d / np.maximum(bound, 1e-300)
    ratio[(d == 0)] = 0.0
    return bool(ratio.max() <= 1.0), {"worst_ratio_to_bound": float(ratio.max()), "fraction_within_bound": float((ratio <= 1.0).mean()),
                                      "max_abs_diff": float(d.max())}


# ---- what the float formula leaves out ------------------------------------------------------------------------------------
# SURVEY 8(d)'s float formula prices the rounding of the tap sums only.  The weights themselves come out of a degree-6 Horner chain
# in x^2 whose terms reach |a6| 4^6 = 3.5 and |a1| 4 = 8.2 while their sum goes to 0 as |x| -> 2: EVERY fp32 evaluation order of
# generators.cpp:31-47 carries an absolute weight error of up to ~1e-6, whatever the weight's size, so two legal orders (and the
# reference's own order against real arithmetic) differ by far more than eps * sum|w v| / |sum w| wherever a large sample sits
# under a near-zero weight.  `lanczos_real` evaluates the same formula -- the same fp32 sampling positions and fractions, the
# polynomial and the sums in float64 -- as the yardstick: a float mode is as good as the reference's rounding order when it is
# no further from this value than that order is (tests/test_warp_gate_gpu.py asserts exactly that, and records by how much the
# formula itself is missed).
def lanczos_real(src, tr, rows=None):
    h, w, c = src.shape
    p = O.ul_params_warp(O.Transform.of(*tr), w, h)
    A1 = np.float32(1.0) + np.float32(p[0])
    B, TX, TY = np.float32(p[1]), np.float32(p[2]), np.float32(p[3])
    r0, r1 = rows if rows else (0, h)
    xs = np.arange(w, dtype=np.float32)[None, :]
    ys = np.arange(r0, r1, dtype=np.float32)[:, None]
    Wx = (A1 * xs - B * ys) + TX
    Wy = (B * xs + A1 * ys) + TY
    flx, fly = np.floor(Wx), np.floor(Wy)
    frx, fry = Wx - flx, Wy - fly
    ix, iy = flx.astype(np.int64), fly.astype(np.int64)
    s = src.astype(np.float64)

    def poly(a32):                                   # the argument is the fp32 value the kernel forms; the rest in float64
        a = a32.astype(np.float64)
        x2 = a * a
        v = np.full_like(x2, 0.000858519)
        for k in (-0.0158853, 0.128693, -0.583468, 1.52229, -2.05238, 0.999861):
            v = np.float64(np.float32(k)) + v * x2
        return np.where(np.abs(a) >= 2.0, 0.0, v)
    wx = [poly(np.float32(u - 2) - frx) for u in range(1, 5)]
    wy = [poly(np.float32(u - 2) - fry) for u in range(1, 5)]
    num = np.zeros((r1 - r0, w, c))
    den = np.zeros((r1 - r0, w))
    for ry in range(4):
        sy = np.clip(iy + ry - 1, 0, h - 1)
        for rx in range(4):
            sx = np.clip(ix + rx - 1, 0, w - 1)
            w2 = wx[rx] * wy[ry]
            num += w2[..., None] * s[sy, sx]
            den += w2
    return num / den[..., None]
