"""Host-side scalar logic of the product (transform algebra, tile rule, smoother) through the C ABI,
against the oracle and against the reference's own property tests (align_test.cpp:261-601)."""
import numpy as np
import pytest


def _rt(rng):
    return (rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(-50, 50), rng.uniform(-50, 50))


def test_defaults_match_reference_headers(vs):
    p = vs.aligner_params()          # alignment.hpp:5-41
    assert (p.phase_correlate, p.phase_correlate_threshold, p.threshold, p.max_iters) == (0, 0.5, 0.02, 64)
    assert p.smallest_fraction == np.float32(0.8) and (p.pyramid_min_width, p.pyramid_min_height) == (20, 20)
    assert p.max_displacement == 10.0
    s = vs.stabilizer_params()       # stabilizer.hpp:13-30
    assert (s.lag, s.smoother_memory, s.lambda_, s.enable_smoother, s.crop_pixels) == (10, 5, 4.0, 1, 32)
    assert (s.min_disp, s.max_disp, s.min_decay, s.max_decay) == (48.0, 64.0, 0.9, 0.7)


def test_transform_algebra_identical_to_oracle(vs, oracle):
    rng = np.random.default_rng(6789)
    for _ in range(200):
        a, b = _rt(rng), _rt(rng)
        ta, tb = vs.Transform.of(*a), vs.Transform.of(*b)
        oa, ob = oracle.Transform.of(*a), oracle.Transform.of(*b)
        assert vs.t_inverse(ta).tup() == oracle.t_inverse(oa).tup()
        assert vs.t_compose(ta, tb).tup() == oracle.t_compose(oa, ob).tup()
        p = (rng.uniform(-500, 500), rng.uniform(-500, 500))
        assert vs.t_warp(ta, *p) == oracle.t_warp(oa, *p)
        assert vs.t_warp(ta, *p, center=(320.0, 180.5)) == oracle.t_warp(oa, *p, center=(320.0, 180.5))
        assert vs.t_max_corner_displacement(ta, 1920, 1080) == oracle.t_max_corner_displacement(oa, 1920, 1080)
        assert np.array_equal(vs.ul_params_sparse(ta, 1920, 1080), oracle.ul_params_sparse(oa, 1920, 1080))
        assert np.array_equal(vs.ul_params_warp(ta, 1919, 1081), oracle.ul_params_warp(oa, 1919, 1081))


def test_reference_property_tests(vs):
    # align_test.cpp:444-601 (TestRandomizedInverse / Compose / InverseComposeIdentity), EPSILON = 1e-5
    rng = np.random.default_rng(9999)
    for _ in range(50):
        t = vs.Transform.of(*_rt(rng))
        ti = vs.t_inverse(t)
        for _ in range(10):
            p = (rng.uniform(-100, 100), rng.uniform(-100, 100))
            u = vs.t_warp(ti, *vs.t_warp(t, *p))
            assert abs(u[0] - p[0]) < 1e-5 and abs(u[1] - p[1]) < 1e-5
        t2 = vs.Transform.of(*_rt(rng))
        t3 = vs.t_compose(t, t2)
        p = (rng.uniform(-100, 100), rng.uniform(-100, 100))
        a, b = vs.t_warp(t3, *p), vs.t_warp(t2, *vs.t_warp(t, *p))
        assert abs(a[0] - b[0]) < 1e-5 and abs(a[1] - b[1]) < 1e-5
        i1 = vs.t_compose(t, ti)
        assert max(abs(v) for v in i1.tup()) < 1e-5


@pytest.mark.parametrize("w,h", [(640, 480), (320, 240), (1920, 1080), (3840, 2160), (480, 270), (120, 67), (60, 33), (17, 9)])
def test_tile_size(vs, oracle, w, h):
    assert vs.tile_size(w, h) == oracle.tile_size(w, h)


def test_tvl1_and_smoother_identical_to_oracle(vs, oracle):
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 16):
        d = rng.normal(scale=5, size=n)
        assert vs.tvl1_smooth(d, 4.0).tolist() == oracle.tvl1_smooth(d, 4.0).tolist()
    g, c = vs.Smoother(10, 5, 4.0), oracle.Smoother(10, 5, 4.0)
    for _ in range(40):
        m = rng.normal(scale=3, size=4)
        okg, tg = g.update(vs.Transform.of(*m))
        okc, tc = c.update(oracle.Transform.of(*m))
        assert okg == okc and tg.tup() == tc.tup()


@pytest.mark.parametrize("lag,memory", [(1, 0), (3, 7), (10, 5), (30, 33), (50, 40)])
def test_smoother_window_sizes(vs, oracle, lag, memory):
    """the four parameters are swept interleaved (windows up to 64 samples) or one after the other (longer windows): both must be
    the oracle's scalar sweep bit for bit, through the warm-up (growing window) and in the steady state, for jumps beyond lambda too"""
    rng = np.random.default_rng(100 * lag + memory)
    g, c = vs.Smoother(lag, memory, 4.0), oracle.Smoother(lag, memory, 4.0)
    for i in range(2 * (lag + memory) + 20):
        m = rng.normal(scale=(0.01, 3.0, 40.0)[i % 3], size=4)
        okg, tg = g.update(vs.Transform.of(*m))
        okc, tc = c.update(oracle.Transform.of(*m))
        assert okg == okc and tg.tup() == tc.tup()


def test_format_bits(vs, oracle):
    """16-bit containers carry their sample depth in the format (luma shift = bits - 8, warp saturation = vs_format_max_value)"""
    want = {vs.FMT_GRAY8: 8, vs.FMT_BGR8: 8, vs.FMT_BGR10: 10, vs.FMT_BGR12: 12, vs.FMT_BGR16_FULL: 16, 5: 0, 6: 0, -1: 0}
    for f, b in want.items():
        assert vs.lib().vs_format_bits(f) == b
        assert oracle.lib().vso_format_bits(f) == b
        assert vs.lib().vs_format_max_value(f) == ((1 << b) - 1 if b else 0)
    # (5 was the first release's VS_FMT_BGR16 -- 10-bit luma, 65535 saturation -- retired with ABI 5: an unknown format on both sides)
    assert vs.lib().vs_abi_version() == 5
