"""The on-device replica of libstdc++'s std::nth_element (alignment.cpp:466-486) must leave the same
elements in the same order as the host's std::nth_element -- the reference's result depends on both."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(0)
    out = []
    for trial in range(160):
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
