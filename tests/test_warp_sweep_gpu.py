"""Seeded random sweep of bgr_image_warp against the CPU restatement, bit for bit: frame sizes from one pixel up,
8 / 10 / 12 / 16-bit containers, the five samplers (the three Lanczos2 forms, the float bilinear, cv::warpAffine's fixed-point bilinear), both borders, transforms from near-identity to degenerate (zero scale, mirror,
far outside the frame), two-frame batches and output windows (a window = the same rows and columns cut out of the whole warp)."""
import os

import numpy as np
import pytest

from _diff import same

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))     # a soak run draws this many times the cases (seeds continue upward)


def _transform(rng):
    kind = int(rng.integers(0, 6))
    if kind == 0:      # stabilisation-sized
        return (rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01), rng.uniform(-8, 8), rng.uniform(-8, 8))
    if kind == 1:      # visible rotation / zoom: still inside the LDS window for most tiles
        return (rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(-30, 30), rng.uniform(-30, 30))
    if kind == 2:      # large: the per-pixel global path
        return (rng.uniform(-0.9, 2.0), rng.uniform(-1.5, 1.5), rng.uniform(-100, 100), rng.uniform(-100, 100))
    if kind == 3:      # everything samples one point (scale 0) or a mirror image (scale -1)
        return (float(rng.choice([-1.0, -2.0])), 0.0, rng.uniform(0, 50), rng.uniform(0, 50))
    if kind == 4:      # far outside the frame: border handling only
        return (0.0, 0.0, float(rng.choice([-1e4, 1e4, 3e5])), float(rng.choice([-1e4, 1e4])))
    return (0.0, 0.0, float(rng.integers(-5, 6)), float(rng.integers(-5, 6)))     # integer shift: weights exactly {0, 1}


@pytest.mark.parametrize("seed", range(200 * _SCALE))
def test_random_warp_is_bit_exact(gpu_vs, oracle, seed):
    rng = np.random.default_rng(31000 + seed)
    small = rng.random() < 0.3
    w = int(rng.integers(1, 70)) if small else int(rng.integers(70, 420))
    h = int(rng.integers(1, 40)) if small else int(rng.integers(40, 260))
    bits = int(rng.choice([8, 8, 10, 12, 16]))
    max_value = (1 << bits) - 1
    mode, border = int(rng.integers(0, 5)), int(rng.integers(0, 2))
    dt = np.uint8 if bits == 8 else np.uint16
    n = int(rng.integers(1, 3))
    src = rng.integers(0, max_value + 1, (n, h, w, 3)).astype(dt)
    trs = [_transform(rng) for _ in range(n)]
    exp = np.stack([oracle.bgr_image_warp(src[i], oracle.Transform.of(*trs[i]), mode, border, max_value=max_value) for i in range(n)])
    ts = [gpu_vs.Transform.of(*t) for t in trs]
    got = gpu_vs.bgr_image_warp_batch(src, ts, mode, border, max_value=max_value)
    assert same(got, exp), (w, h, bits, mode, border, trs)
    # a window of the same warp
    rw, rh = int(rng.integers(1, w + 1)), int(rng.integers(1, h + 1))
    rx, ry = int(rng.integers(0, w - rw + 1)), int(rng.integers(0, h - rh + 1))
    win = gpu_vs.bgr_image_warp_roi_batch(src, ts, (rx, ry, rw, rh), mode, border, max_value=max_value)
    assert same(win, exp[:, ry:ry + rh, rx:rx + rw]), (w, h, bits, mode, border, trs, (rx, ry, rw, rh))


@pytest.mark.parametrize("seed", range(80 * _SCALE))
def test_random_generic_warp_forms_are_bit_exact(gpu_vs, oracle, seed):
    """what the tuned 3-channel kernel does not take: 1 / 2 / 4 channels, float output (the type of the reference's image_warp, generators.cpp:126-164),
    image_warp itself (gray u8 -> f32 bilinear) and the BGR -> gray conversion, on random sizes and transforms"""
    rng = np.random.default_rng(41000 + seed)
    small = rng.random() < 0.3
    w = int(rng.integers(1, 40)) if small else int(rng.integers(40, 300))
    h = int(rng.integers(1, 30)) if small else int(rng.integers(30, 200))
    tr = _transform(rng)
    mode, border = int(rng.integers(0, 5)), int(rng.integers(0, 2))
    c = int(rng.choice([1, 2, 3, 4]))
    bits = int(rng.choice([8, 16]))
    src = rng.integers(0, 256 if bits == 8 else 65536, (h, w, c)).astype(np.uint8 if bits == 8 else np.uint16)
    f32 = bool(rng.integers(0, 2)) or c == 3                     # (3 channels with integer output is the tuned kernel's job: the other test)
    if mode == gpu_vs.WARP_BILINEAR_CV:
        f32 = False                                              # (cv::warpAffine's integer path has no float output)
    g = gpu_vs.bgr_image_warp(src, gpu_vs.Transform.of(*tr), mode, border, f32=f32)
    o = oracle.bgr_image_warp(src, oracle.Transform.of(*tr), mode, border, f32=f32)
    assert same(g, o, equal_nan=True), (w, h, c, bits, mode, border, f32, tr)
    gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
    assert same(gpu_vs.image_warp(gray, gpu_vs.Transform.of(*tr)), oracle.image_warp(gray, oracle.Transform.of(*tr)), equal_nan=True), (w, h, tr)
    bgr = rng.integers(0, 256 if bits == 8 else 1024, (h, w, 3)).astype(src.dtype)
    assert same(gpu_vs.bgr_to_gray(bgr), oracle.bgr_to_gray(bgr)), (w, h, bits)


@pytest.mark.parametrize("seed", range(100 * _SCALE))
def test_random_pitches_and_unaligned_device_pointers(gpu_vs, oracle, seed):
    """the same warps through buffers as callers really hold them: rows longer than the image (pitches that leave every other row unaligned),
    device pointers that do not start on a dword, frame strides with slack -- host and device memory.  The aligned 12-byte loads and
    3-dword stores of the tuned kernel must give way to the byte paths exactly where they have to."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(45000 + seed)
    w, h = int(rng.integers(1, 330)), int(rng.integers(1, 120))
    bits = int(rng.choice([8, 8, 10]))
    mode, border = int(rng.integers(0, 5)), int(rng.integers(0, 2))
    n = int(rng.integers(1, 4))
    max_value = 255 if bits == 8 else 1023
    dt = np.uint8 if bits == 8 else np.uint16
    esz = 1 if bits == 8 else 2
    frames = rng.integers(0, max_value + 1, (n, h, w, 3)).astype(dt)
    trs = [_transform(rng) for _ in range(n)]
    exp = np.stack([oracle.bgr_image_warp(frames[i], oracle.Transform.of(*trs[i]), mode, border, max_value=max_value) for i in range(n)])
    sp, dp = 3 * w + int(rng.integers(0, 8)), 3 * w + int(rng.integers(0, 8))          # row pitches in elements
    sfs, dfs = h * sp + int(rng.integers(0, 5)), h * dp + int(rng.integers(0, 5))      # frame strides in elements
    so, do = int(rng.integers(0, 4)), int(rng.integers(0, 4))                          # the first frame starts this many elements into its buffer
    src = rng.integers(0, max_value + 1, so + n * sfs + 8).astype(dt)                  # (garbage everywhere the image is not)
    for i in range(n):
        rows = np.lib.stride_tricks.as_strided(src[so + i * sfs:], (h, 3 * w), (sp * esz, esz))
        rows[...] = frames[i].reshape(h, 3 * w)
    dst_fill = int(rng.integers(0, max_value + 1))
    dst = np.full(do + n * dfs + 8, dst_fill, dt)
    arr = (gpu_vs.Transform * n)(*[gpu_vs.Transform.of(*t) for t in trs])
    L = gpu_vs.lib()
    if rng.random() < 0.5:
        r = L.vs_bgr_image_warp_batch(C.c_void_p(src.ctypes.data + so * esz), sfs, n, w, h, sp, 3, 8 * esz, arr, mode, border, max_value,
                                      C.c_void_p(dst.ctypes.data + do * esz), dfs, dp, gpu_vs.MEM_HOST, None)
        assert r >= 0, gpu_vs.lib().vs_last_error()
        got = dst
    else:
        ds = torch.from_numpy(src.view(np.int16) if esz == 2 else src).to("cuda:0")
        dd = torch.from_numpy(dst.view(np.int16) if esz == 2 else dst).to("cuda:0")
        r = L.vs_bgr_image_warp_batch(C.c_void_p(ds.data_ptr() + so * esz), sfs, n, w, h, sp, 3, 8 * esz, arr, mode, border, max_value,
                                      C.c_void_p(dd.data_ptr() + do * esz), dfs, dp, gpu_vs.MEM_DEVICE, None)
        assert r >= 0, gpu_vs.lib().vs_last_error()
        torch.cuda.synchronize()
        got = dd.cpu().numpy()
        got = got.view(np.uint16) if esz == 2 else got
    keep = np.ones(got.shape, bool)

    def again():
        """(only on a mismatch) the same host call once more into a fresh destination: says whether the wrong result is repeatable"""
        d2 = np.full(do + n * dfs + 8, dst_fill, dt)
        L.vs_bgr_image_warp_batch(C.c_void_p(src.ctypes.data + so * esz), sfs, n, w, h, sp, 3, 8 * esz, arr, mode, border, max_value,
                                  C.c_void_p(d2.ctypes.data + do * esz), dfs, dp, gpu_vs.MEM_HOST, None)
        return "a second identical call %s" % ("gives the SAME output" if np.array_equal(d2, got) else "gives a DIFFERENT output (transient)")

    for i in range(n):
        rows = np.lib.stride_tricks.as_strided(got[do + i * dfs:], (h, 3 * w), (dp * esz, esz))
        d = same(rows, exp[i].reshape(h, 3 * w))
        assert d, (d, "frame %d" % i, w, h, bits, mode, border, sp, dp, so, do, trs[i], "dst_fill %d" % dst_fill, "host" if got is dst else "device", again())
        np.lib.stride_tricks.as_strided(keep[do + i * dfs:], (h, 3 * w), (dp, 1))[...] = False
    outside = np.flatnonzero(keep & (got != dst_fill))
    assert outside.size == 0, "%d elements outside the output rows were written; first: element %d = %d (fill %d)" % (outside.size, outside[0], got[outside[0]], dst_fill)


@pytest.mark.parametrize("seed", range(30 * _SCALE))
def test_non_finite_transforms_stay_inside_their_buffers(gpu_vs, seed):
    """A transform with NaN / infinite components has no meaningful result (the reference's cast<int>(floor(W)) is undefined there) -- but the call must
    return, must not touch a byte outside its output rows, and (on the bounds-checked build, where this file also runs) must not index outside its
    tile.  Every kernel form: the tuned 3-channel kernels in their five modes and both depths, the generic kernel, float output, image_warp."""
    import ctypes as C
    rng = np.random.default_rng(47000 + seed)
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 90))
    bad = [float("nan"), float("inf"), float("-inf"), 3.0e38, -3.0e38, 1e20]
    tr = [float(rng.choice(bad)) if rng.random() < 0.6 else float(rng.uniform(-1, 1)) for _ in range(4)]
    if all(np.isfinite(v) and abs(v) < 1e10 for v in tr):
        tr[int(rng.integers(0, 4))] = float("nan")
    t = gpu_vs.Transform.of(*tr)
    for bits in (8, 16):
        dt = np.uint8 if bits == 8 else np.uint16
        for c in (1, 3):
            src = rng.integers(0, 256, (h, w, c)).astype(dt)
            for mode in range(5):
                for border in (0, 1):
                    pad = 5
                    dst = np.full((h, w * c + pad), 77, dt)
                    r = gpu_vs.lib().vs_bgr_image_warp(C.c_void_p(src.ctypes.data), w, h, w * c, c, bits, C.byref(t), mode, border, 255 if bits == 8 else 1023,
                                                       C.c_void_p(dst.ctypes.data), w * c + pad, gpu_vs.MEM_HOST, None)
                    assert r >= 0, gpu_vs.lib().vs_last_error()
                    assert np.all(dst[:, w * c:] == 77), (w, h, c, bits, mode, border, tr)
            gpu_vs.bgr_image_warp(src, t, 0, 0, f32=True)
    gpu_vs.image_warp(rng.integers(0, 256, (h, w), dtype=np.uint8), t)
