"""Array comparison for the parity tests: `same(got, want)` is np.array_equal that SAYS WHERE the arrays differ.

Round 5's one red run printed two truncated arrays and no coordinate; every bit-exact assert of the sweeps now reports the shapes, how many
elements differ, the first differing index with both values, the bounding box of the differences and the set of wrong values (if small),
so that a failure seen once on somebody else's box can be classified from the log alone (tools/repro_warp_batch.py, profiles/r06_flake.md)."""
import numpy as np


class _Diff:
    """truthy when equal; its repr is the explanation pytest shows for `assert same(...)`"""

    def __init__(self, ok, text):
        self.ok, self.text = ok, text

    def __bool__(self):
        return self.ok

    def __repr__(self):
        return self.text


def same(got, want, equal_nan=False):
    g, w = np.asarray(got), np.asarray(want)
    if g.shape != w.shape:
        return _Diff(False, "shapes differ: got %r want %r" % (g.shape, w.shape))
    ne = g != w
    if equal_nan and g.dtype.kind == "f":
        ne &= ~(np.isnan(g) & np.isnan(w))
    n = int(np.count_nonzero(ne))
    if n == 0:
        return _Diff(True, "equal")
    idx = np.argwhere(ne)
    first = tuple(int(v) for v in idx[0])
    lo, hi = idx.min(0).tolist(), idx.max(0).tolist()
    wrong = np.unique(g[ne])
    vals = wrong.tolist() if wrong.size <= 8 else "%d distinct values, e.g. %r" % (wrong.size, wrong[:8].tolist())
    return _Diff(False, "%d of %d elements differ (shape %r, dtype %s); first at index %r: got %r want %r; differing indices span %r..%r; wrong values: %s"
                 % (n, g.size, g.shape, g.dtype, first, g[first].item(), w[first].item(), lo, hi, vals))
