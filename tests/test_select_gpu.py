"""The on-device replica of libstdc++'s std::nth_element (alignment.cpp:466-486) must leave the same
elements in the same order as the host's std::nth_element -- the reference's result depends on both."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))     # soak runs: this many times the random cases


def _cases():
    rng = np.random.default_rng(0)
    out = []
    for trial in range(160 * _SCALE):
        tx, ty = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        hi = int(rng.choice([1, 2, 3, 5, 20, 300, 60000]))
        wd = rng.integers(0, hi, (ty, tx)).astype(np.uint16)
        if trial % 7 == 0:
            wd[:] = 5
        if trial % 11 == 0:
            wd = np.sort(wd.ravel()).reshape(ty, tx)
        if trial % 13 == 0:
            wd = np.sort(wd.ravel())[::-1].reshape(ty, tx).copy()
        out.append((wd, float(rng.choice([0.8, 0.5, 0.99, 0.1, 1.0]))))
    return out


def test_select_matches_std_nth_element(gpu_vs, oracle):
    for wd, frac in _cases():
        got, status = gpu_vs.select_smallest(wd, frac)
        ref = oracle.select_smallest(wd, frac)
        assert status[0] == 0
        assert np.array_equal(got[0], ref), (wd.shape, frac)
        # the documented STL-independent rule on the same tables (oracle select rule 1)
        assert np.array_equal(gpu_vs.select_smallest_stable(wd, frac)[0], oracle.select_smallest_stable(wd, frac)), (wd.shape, frac)


@pytest.mark.parametrize("tx,ty", [(40, 30), (96, 54), (192, 108), (45, 44), (2, 2), (3, 1), (1, 1), (161, 161)])
def test_select_reference_shapes_batched(gpu_vs, oracle, tx, ty):
    # the tile grids of BASELINE's configs (SURVEY 8d shape table) with warpdiff-like value distributions
    rng = np.random.default_rng(tx * 1000 + ty)
    wd = np.minimum(rng.poisson(3.0, (6, ty, tx)), 65535).astype(np.uint16)
    wd[1] = rng.integers(0, 2, (ty, tx))
    wd[2] = 0
    wd[3] = rng.integers(0, 65536, (ty, tx))
    got, status = gpu_vs.select_smallest(wd, 0.8)
    assert not status.any()
    for i in range(6):
        assert np.array_equal(got[i], oracle.select_smallest(wd[i], 0.8)), i


def test_select_capacity_is_an_error(gpu_vs):
    with pytest.raises(gpu_vs.VsError):
        gpu_vs.select_smallest(np.zeros((200, 200), np.uint16))


@pytest.mark.parametrize("tx,ty", [(40, 30), (96, 54), (45, 44), (48, 40), (161, 161), (20, 13)])
def test_median_of_3_killer_is_flagged(gpu_vs, oracle, tx, ty):
    """libstdc++'s introselect gives up after 2 lg n partition rounds and heap-selects (bits/stl_algo.h); the on-device replica
    does not restate __heap_select -- it must FLAG such a table (status 1) so that the engine sends the pair through the host's
    std::nth_element.  The table comes from McIlroy's adversary run against the host's own std::nth_element (oracle)."""
    wd = oracle.nth_element_killer(tx, ty)
    assert oracle.nth_element_hits_depth_limit(wd)
    # the host call copes (heap-select) and still returns the smallest 80 %
    ref = oracle.select_smallest(wd)
    n = len(ref)
    assert np.array_equal(np.sort(wd.ravel()[ref]), np.sort(wd.ravel())[:n])
    # batch of three: the killer between two ordinary tables -- only the killer is flagged, the others are unaffected
    rng = np.random.default_rng(tx + ty)
    batch = np.stack([rng.integers(0, 300, (ty, tx)).astype(np.uint16), wd, np.minimum(rng.poisson(3.0, (ty, tx)), 65535).astype(np.uint16)])
    got, status = gpu_vs.select_smallest(batch, 0.8)
    assert list(status) == [0, 1, 0]
    for i in (0, 2):
        assert not oracle.nth_element_hits_depth_limit(batch[i])
        assert np.array_equal(got[i], oracle.select_smallest(batch[i])), i


def test_depth_limit_flag_agrees_with_the_restated_control_flow(gpu_vs, oracle):
    """status == what a literal restatement of __introselect's control flow says, on killers cut short and padded: tables that
    stop one round before the budget runs out are NOT flagged and select like the host"""
    flagged = 0
    for (tx, ty) in [(40, 30), (64, 33), (45, 44)]:
        k = oracle.nth_element_killer(tx, ty).ravel()
        for keep in (len(k), len(k) * 3 // 4, len(k) // 2, len(k) // 4):
            wd = k.copy()
            wd[keep:] = wd[:keep].max() + 1          # the tail of the adversary's values replaced by one large value
            wd = wd.reshape(ty, tx)
            want = oracle.nth_element_hits_depth_limit(wd)
            got, status = gpu_vs.select_smallest(wd, 0.8)
            assert bool(status[0]) == want, (tx, ty, keep)
            flagged += int(want)
            if not want:
                assert np.array_equal(got[0], oracle.select_smallest(wd))
    assert flagged >= 3
