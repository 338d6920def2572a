"""Seeded random sweep of AlignNextFrame against the CPU oracle: frame size, channel count, bit depth, aligner parameters
(alignment.hpp:9-21 VideoAlignerParams), selection mode and the camera path's roughness all drawn at random.  Same gate as
test_engine_gpu.py (its _check_seq): success / failure decisions, failure reasons, iteration counts per level, selected-point
counts, condition numbers and the per-level and final transforms (1e-4)."""
import os

import numpy as np
import pytest

from test_engine_gpu import TOL, _check_seq, _cmp_transform, _run_both

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))     # a soak run draws this many times the cases (seeds continue upward)


def _draw(rng):
    w = int(rng.integers(97, 700))
    h = int(rng.integers(65, 460))
    # at least three pyramid levels (alignment.cpp:185-195: halve while both extents stay >= the minimum), else the call is an error
    min_w = int(rng.integers(8, max(9, w // 4)))
    min_h = int(rng.integers(8, max(9, h // 4)))
    params = dict(pyramid_min_width=min_w, pyramid_min_height=min_h,
                  smallest_fraction=float(rng.choice([0.5, 0.65, 0.8, 0.9, 1.0])),
                  max_iters=int(rng.choice([3, 8, 24, 64])),
                  threshold=float(rng.choice([0.01, 0.05, 0.2])),
                  max_displacement=float(rng.choice([4.0, 16.0, 64.0])),
                  phase_correlate=int(rng.integers(0, 2)))
    return w, h, int(rng.choice([1, 3])), int(rng.choice([8, 8, 10])), params, float(rng.choice([0.5, 2.0, 6.0]))


@pytest.mark.parametrize("seed", range(96 * _SCALE))
def test_random_configuration_matches_the_oracle(gpu_vs, oracle, seed):
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(9000 + seed)
    w, h, ch, bits, params, rough = _draw(rng)
    if params["phase_correlate"]:
        # (the FFT of pyramid level 2 takes power-of-two-free sizes too, but the oracle's DFT is O(n^2) per row: keep those frames small)
        w, h = min(w, 360), min(h, 260)
        params["pyramid_min_width"] = min(params["pyramid_min_width"], w // 4)
        params["pyramid_min_height"] = min(params["pyramid_min_height"], h // 4)
    frames, _ = synth.make_clip(w, h, 5, seed=700 + seed, channels=ch, bits=bits, jitter_t=rough)
    if bits == 10 and ch == 1:
        frames = (frames >> 2).astype(np.uint8)            # (gray frames are 8-bit only: VS_FMT_GRAY8)
    mode = int(rng.choice([gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE, gpu_vs.SELECT_STABLE]))     # (stable: against the oracle's rule 1)
    gpu, cpu, res = _run_both(gpu_vs, oracle, frames, select_mode=mode, **params)
    _check_seq(res)
    assert not res[0][0]                                    # first frame: no previous frame (alignment.cpp:231-234)


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_random_stabilizer_configuration_matches_the_oracle(gpu_vs, oracle, seed):
    """processFrame (stabilizer.cpp:9-117) with random VideoStabilizerParams (stabilizer.hpp:13-30) and both warp modes / borders:
    same has-output sequence, same measurement / accumulated transforms, frames within 1 LSB (the gate of test_stabilizer_matches_oracle)."""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(12000 + seed)
    w, h = int(rng.integers(160, 520)), int(rng.integers(120, 360))
    bits = int(rng.choice([8, 8, 10]))
    crop = int(rng.integers(0, min(w, h) // 5))
    kw = dict(lag=int(rng.integers(1, 7)), smoother_memory=int(rng.integers(0, 6)), crop_pixels=crop,
              enable_smoother=int(rng.integers(0, 2)), warp_mode=int(rng.choice([gpu_vs.WARP_LANCZOS2, gpu_vs.WARP_BILINEAR, gpu_vs.WARP_LANCZOS2_FAST])),
              warp_border=int(rng.integers(0, 2)), min_disp=float(rng.choice([0.5, 2.0])), max_disp=float(rng.choice([8.0, 40.0])))
    kw["lambda"] = float(rng.choice([0.5, 2.0, 8.0]))
    n = 14
    frames, _ = synth.make_clip(w, h, n, seed=300 + seed, channels=3, bits=bits, jitter_t=float(rng.choice([1.0, 4.0, 12.0])))
    if rng.random() < 0.33:                                 # the documented STL-independent selection rule on both sides
        g, c = gpu_vs.Stabilizer(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw), oracle.Stabilizer(select_rule=oracle.SELECT_STABLE, **kw)
    else:
        g, c = gpu_vs.Stabilizer(device=0, **kw), oracle.Stabilizer(**kw)
    produced = 0
    for i, f in enumerate(frames):
        og, oc = g.process(f), c.process(f)
        assert (og is None) == (oc is None), i
        mg, ag, sg = g.state()
        mc, ac, sc = c.state()
        assert sg == sc, i
        assert _cmp_transform(mg, mc) < TOL and _cmp_transform(ag, ac) < 10 * TOL, i
        if oc is not None:
            produced += 1
            assert og.shape == oc.shape == (h - 2 * crop, w - 2 * crop, 3) and og.dtype == oc.dtype
            d = np.abs(og.astype(np.int32) - oc.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < 1e-2, (i, int(d.max()), float((d != 0).mean()))
    assert produced == n - kw["lag"]
