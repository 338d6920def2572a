"""Seeded random sweep of the keyframe / template chain at kernel level, bit for bit against the CPU restatement: pyr_down (a2), grad_xy
(a3), the fused keyframe tables = grad_argmax + sparse_jac (a4, a5) with the reference's tile-size rule or a forced tile size,
sparse_warpdiff (a6) and sparse_ica (a9, 1e-12 relative: fp64 tree sum against the serial sum) -- on random frame sizes (odd, tiny,
wider than a strip), content from smooth to tie-heavy, and random transforms."""
import os

import numpy as np
import pytest

from _diff import same

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))     # a soak run draws this many times the cases (seeds continue upward)


def _content(rng, w, h):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 1:      # few levels: ties everywhere
        return (rng.integers(0, 4, (h, w)) * 85).astype(np.uint8)
    if kind == 2:      # smooth ramp + noise
        y, x = np.mgrid[0:h, 0:w]
        return ((x * 3 + y * 2 + rng.integers(0, 6, (h, w))) % 256).astype(np.uint8)
    img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)                  # flat with a few blobs
    for _ in range(int(rng.integers(0, 6))):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        img[y0:y0 + int(rng.integers(1, 9)), x0:x0 + int(rng.integers(1, 9))] = int(rng.integers(0, 256))
    return img


@pytest.mark.parametrize("seed", range(60 * _SCALE))
def test_random_chain_is_bit_exact(gpu_vs, oracle, seed):
    rng = np.random.default_rng(52000 + seed)
    w, h = int(rng.integers(8, 900)), int(rng.integers(8, 420))
    key = _content(rng, w, h)
    assert same(gpu_vs.pyr_down(key), oracle.pyr_down(key))
    gx, gy = gpu_vs.grad_xy(key)
    ogx, ogy = oracle.grad_xy(key)
    assert same(gx, ogx) and same(gy, ogy)
    ts = None if rng.random() < 0.5 else int(rng.integers(2, min(w, h, 40) + 1))
    gts, lmx, lmy, jx, jy = gpu_vs.keyframe_fused(key, ts=ts)
    ots, olx, oly = oracle.grad_argmax(ogx, ogy, ts=ts)
    ojx, ojy = oracle.sparse_jac(ogx, ogy, olx, oly)
    assert gts == ots
    assert same(lmx, olx) and same(lmy, oly)
    assert same(jx, ojx) and same(jy, ojy)
    # a template = the keyframe shifted a little, plus noise; a transform near the truth or far from it
    tmpl = np.roll(key, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), (0, 1))
    tmpl = np.clip(tmpl.astype(np.int16) + rng.integers(-3, 4, tmpl.shape), 0, 255).astype(np.uint8)
    tr = (rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), rng.uniform(-6, 6), rng.uniform(-6, 6)) if rng.random() < 0.8 else \
         (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-300, 300), rng.uniform(-300, 300))
    for lm in (olx, oly):
        g = gpu_vs.sparse_warpdiff(tmpl, key, lm, gpu_vs.Transform.of(*tr))
        o = oracle.sparse_warpdiff(tmpl, key, lm, oracle.Transform.of(*tr))
        assert same(g, o), (w, h, tr)
    nt = olx.shape[1] * olx.shape[2]
    n = max(1, int(nt * 0.8))
    kx, ky = rng.permutation(nt)[:n], rng.permutation(nt)[:n]
    args = (olx.reshape(2, -1)[:, kx], oly.reshape(2, -1)[:, ky], ojx.reshape(4, -1)[:, kx], ojy.reshape(4, -1)[:, ky])
    g = gpu_vs.sparse_ica(tmpl, key, *args, gpu_vs.Transform.of(*tr))
    o = oracle.sparse_ica(tmpl, key, *args, oracle.Transform.of(*tr))
    assert np.abs(g - o).max() <= 1e-12 * (np.abs(o).max() + 1e-30) + 1e-9, (w, h, tr, g, o)
