"""Seeded random sweep of vs_phase_correlate (alignment.cpp:369-388: cv::phaseCorrelate on pyramid level 2, restated in oracle/vs_phase.cpp) against
the oracle, bit for bit: random sizes (so every radix mix of the padded DFT sizes), random content and shifts, flat and tiny images."""
import os

import numpy as np
import pytest

from _diff import same

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))


@pytest.mark.parametrize("seed", range(40 * _SCALE))
def test_random_phase_correlation_is_bit_exact(gpu_vs, oracle, seed):
    rng = np.random.default_rng(93000 + seed)
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 420))
    kind = int(rng.integers(0, 4))
    big = rng.integers(0, 256, (h + 24, w + 24)).astype(np.uint8)
    if kind == 1:                                             # smooth content: a broad peak
        y, x = np.mgrid[0:h + 24, 0:w + 24]
        big = ((np.sin(x * 0.11) + np.cos(y * 0.07)) * 60 + 128 + rng.integers(0, 4, big.shape)).astype(np.uint8)
    dy, dx = int(rng.integers(-8, 9)), int(rng.integers(-8, 9))
    a = np.ascontiguousarray(big[12:12 + h, 12:12 + w])
    b = np.ascontiguousarray(big[12 + dy:12 + dy + h, 12 + dx:12 + dx + w])
    if kind == 2:
        b = a.copy()                                          # identical images: the peak at the origin
    if kind == 3:
        a[:] = int(rng.integers(0, 256)); b[:] = int(rng.integers(0, 256))     # flat: no peak at all
    gx, gy, gr, surf = gpu_vs.phase_correlate(a, b, want_surface=True)
    want = oracle.phase_surface(a.astype(np.float32), b.astype(np.float32))
    assert surf.shape == want.shape
    assert same(surf, want, equal_nan=True), (h, w, kind)
    ox, oy, orr = oracle.phase_correlate(a, b)
    assert (gx, gy, gr) == (ox, oy, orr) or (np.isnan(orr) and np.isnan(gr)), (h, w, kind, (gx, gy, gr), (ox, oy, orr))
