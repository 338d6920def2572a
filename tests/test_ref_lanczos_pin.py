"""SURVEY 8(a) row a1 pinned by the reference's OWN code -- the one piece of the reference this image can build.

/root/reference/lanczos2_opt.cpp includes nothing but the C++ standard library (lanczos2_opt.cpp:1-6).  `make -C oracle
ref-lanczos` compiles it unmodified, where it lies, into oracle/_ref/lanczos2_opt (no stand-ins, no copied source).  The program
re-derives the even polynomial by a least-squares fit against sinc(x) sinc(x/2) (its `fit_even_polynomial_lanczos2`), prints
the fitted a0..a6 and the max / average error of the hard-coded polynomial on [-2, 2] (lanczos2_opt.cpp:318-356).

Pinned here, by running that binary:
  * the coefficients the reference's fit PRODUCES (to the six digits it prints) are the literals of generators.cpp:31-47 as the
    oracle restates them: a float32 Horner built from the printed strings equals vso_lanczos2 bit for bit on a dense grid;
  * the oracle's own error against the reference's baseline_lanczos2 definition, on the reference's grid (step 1e-4),
    reproduces the printed "Max error" and "Avg error".
This pins a1 only.  Everything else on the path needs Halide / OpenCV: row (c) of SURVEY 8 stays "parity unpinned".
"""
import math
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "lanczos2_opt")


@pytest.fixture(scope="module")
def ref_output():
    if not os.path.exists(BIN):
        # build container: the recipe compiles the reference's file in place; GPU box: the prebuilt binary travels with the tree
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref-lanczos"], check=False)
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/lanczos2_opt is not built and the reference checkout is not here to build it from")
    out = subprocess.run([BIN], check=True, capture_output=True, text=True, timeout=120).stdout
    coeffs = {int(k): v for k, v in re.findall(r"^\s*a(\d+)\s*=\s*(\S+)\s*$", out, re.M)}
    assert sorted(coeffs) == list(range(7)), out
    mx = re.search(r"Max error:\s*(\S+)", out).group(1)
    av = re.search(r"Avg error:\s*(\S+)", out).group(1)
    return [coeffs[i] for i in range(7)], float(mx), float(av), out


def test_reference_fit_reproduces_the_published_coefficients(ref_output):
    """the numbers the program prints today = the ones in its own trailing comment (lanczos2_opt.cpp:370-377) = generators.cpp:33-44"""
    coeffs, _, _, _ = ref_output
    assert coeffs == ["0.999861", "-2.05238", "1.52229", "-0.583468", "0.128693", "-0.0158853", "0.000858519"]


def test_oracle_polynomial_is_the_reference_fit(oracle, ref_output):
    coeffs, _, _, _ = ref_output
    a = [np.float32(c) for c in coeffs]                  # float32("0.999861") is the literal 0.999861f
    xs = np.concatenate([np.linspace(-2.5, 2.5, 4001), np.arange(5) - 2 - 0.25, np.arange(5) - 2 - 0.5, [0.0, 2.0, -2.0]]).astype(np.float32)
    for x in xs:
        x2 = np.float32(x * x)
        v = a[6]
        for k in (5, 4, 3, 2, 1, 0):                     # generators.cpp:33-44: val = a_k + val * x2, highest first, fp32
            v = np.float32(a[k] + np.float32(v * x2))
        want = np.float32(0.0) if abs(float(x)) >= 2.0 else v
        assert oracle.lanczos2(float(x)) == want, (x, oracle.lanczos2(float(x)), want)


def test_oracle_error_against_the_reference_baseline_matches_the_printed_figures(oracle, ref_output):
    """lanczos2_opt.cpp:318-333: for (x = -2; x <= 2; x += 1e-4) err = |poly(x) - sinc(x) sinc(x/2)| -- the same loop, the same
    accumulating x, with the oracle's fp32 polynomial in place of the program's double one (they differ by ~1e-7)."""
    _, mx_ref, av_ref, _ = ref_output

    def baseline(x):                                     # lanczos2_opt.cpp:12-27
        if abs(x) >= 2.0:
            return 0.0
        s = lambda z: 1.0 if z == 0.0 else math.sin(math.pi * z) / (math.pi * z)
        return s(x) * s(0.5 * x)

    mx, sm, n, x = 0.0, 0.0, 0, -2.0
    while x <= 2.0:
        # the program evaluates its polynomial for every x of the loop, |x| = 2 included, without the cut-off (:276-289)
        # (the oracle cuts off at |x| >= 2 as generators.cpp:46 does: evaluate one fp32 step inside instead)
        xe = math.copysign(float(np.nextafter(np.float32(2.0), np.float32(0.0))), x) if abs(np.float32(x)) >= 2.0 else x
        e = abs(float(oracle.lanczos2(xe)) - baseline(x))
        mx, sm, n = max(mx, e), sm + e, n + 1
        x += 1e-4
    assert abs(mx - mx_ref) <= 2e-6, (mx, mx_ref)        # printed with 6 significant digits: 0.000383624
    assert abs(sm / n - av_ref) <= 2e-6, (sm / n, av_ref)
    assert mx_ref <= 3.84e-4                             # SURVEY 8(c) known answer 4
