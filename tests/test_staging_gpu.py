"""The staging pool behind the VS_MEM_HOST form of the kernel-level calls (vs_capi.hip, round 6): device block + pinned host mirror, leased per argument,
one dense copy each way, rows scattered by the CPU.  (i) every allocation of a first call can fail and the pool stays usable: the call reports the
failure, the next call is right; (ii) threads lease and release concurrently with arguments of many sizes and every result equals the CPU restatement;
(iii) a pitched output leaves the caller's bytes between the rows alone for every alignment of the destination."""
import os
import subprocess
import sys

import numpy as np
import pytest

from _diff import same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_allocation_of_a_first_staged_call_can_fail():
    """vs_test_fail_alloc(k) walked over the allocations of a process's FIRST bgr_image_warp call on host memory (empty pool, no rings yet: a fresh process
    per k) -- staging blocks (device + pinned, input and output), the parameter / table rings -- until k no longer fires: the call reports the failure,
    the next call in the same process is right"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from video_stabilizer_amd import capi\n"
            "from oracle import oracle as O\n"
            "k, which = int(sys.argv[1]), int(sys.argv[2])\n"
            "mode, omode, border = [(capi.WARP_BILINEAR_CV, O.WARP_BILINEAR_CV, 1), (capi.WARP_LANCZOS2_SEP, O.WARP_LANCZOS2_SEPARABLE, 0)][which]\n"
            "rng = np.random.default_rng(3)\n"
            "src = rng.integers(0, 256, (70, 150, 3), dtype=np.uint8)\n"
            "tr = (0.003, -0.002, 1.25, -0.75)\n"
            "want = O.bgr_image_warp(src, O.Transform.of(*tr), omode, border=border)\n"
            "capi.device_count()\n"
            "capi.test_fail_alloc(k)\n"
            "try:\n"
            "    got = capi.bgr_image_warp(src, capi.Transform.of(*tr), mode=mode, border=border)\n"
            "    assert np.array_equal(got, want)\n"
            "    print('NOFAIL', capi.test_fail_alloc(0))\n"
            "except capi.VsError as e:\n"
            "    assert 'error -2' in str(e), str(e)\n"
            "    print('FAILED', capi.test_fail_alloc(0))\n"
            "assert np.array_equal(capi.bgr_image_warp(src, capi.Transform.of(*tr), mode=mode, border=border), want)\n"
            "assert np.array_equal(capi.bgr_image_warp(src[:40, :77].copy(), capi.Transform.of(*tr), mode=mode, border=border),\n"
            "                      O.bgr_image_warp(src[:40, :77].copy(), O.Transform.of(*tr), omode, border=border))\n"
            "print('THEN OK')\n") % ROOT
    fired = []
    for which in (0, 1):
        k = 1
        while True:
            out = subprocess.run([sys.executable, "-c", code, str(k), str(which)], capture_output=True, text=True, timeout=300)
            assert out.returncode == 0 and "THEN OK" in out.stdout, (which, k, out.stdout[-500:], out.stderr[-2000:])
            if "NOFAIL" in out.stdout:
                break
            k += 1
            assert k < 30, "the walk does not terminate"
        fired.append(k - 1)
    # 2 staging blocks x (device + pinned) + the table ring (fixed-point bilinear) / the parameter ring's device half (Lanczos2: a small call's parameters
    # travel as kernel arguments, so the ring's pinned half is not made)
    assert fired[0] >= 5 and fired[1] >= 5, fired


def test_threads_lease_and_release_concurrently(gpu_vs, oracle):
    import threading
    vs = gpu_vs
    rng = np.random.default_rng(11)
    cases = []
    for i in range(24):
        w, h = int(rng.integers(8, 700)), int(rng.integers(8, 300))
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        tr = (float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)))
        mode = [vs.WARP_BILINEAR_CV, vs.WARP_BILINEAR, vs.WARP_LANCZOS2_SEP][i % 3]
        omode = [oracle.WARP_BILINEAR_CV, oracle.WARP_BILINEAR, oracle.WARP_LANCZOS2_SEPARABLE][i % 3]
        cases.append((src, tr, mode, oracle.bgr_image_warp(src, oracle.Transform.of(*tr), omode, border=1)))
    gray = rng.integers(0, 256, (200, 320), dtype=np.uint8)
    want_pyr = oracle.pyr_down(gray)
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for j in range(t, len(cases), 4):
                    src, tr, mode, want = cases[j]
                    got = vs.bgr_image_warp(src, vs.Transform.of(*tr), mode=mode, border=1)
                    d = same(got, want)
                    if not d:
                        errors.append((t, rep, j, repr(d)))
                if not np.array_equal(vs.pyr_down(gray), want_pyr):
                    errors.append((t, rep, "pyr_down"))
        except Exception as e:      # noqa: BLE001 -- reported below
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:4]


@pytest.mark.parametrize("offset", [0, 1, 2, 3])
def test_pitched_output_at_every_alignment_keeps_the_callers_bytes(gpu_vs, oracle, offset):
    """a 10-bit 2-frame batch into a destination that starts `offset` ELEMENTS into its buffer (2-byte steps: every alignment of a dword) with rows longer
    than the image: the rows equal the twin, every byte between and around them is still the fill"""
    import ctypes as C
    vs = gpu_vs
    rng = np.random.default_rng(20 + offset)
    w, h, n = 101, 37, 2
    frames = rng.integers(0, 1024, (n, h, w, 3)).astype(np.uint16)
    trs = [(0.002, 0.001, 0.5, -1.25), (-0.004, 0.003, -2.0, 0.75)]
    exp = np.stack([oracle.bgr_image_warp(frames[i], oracle.Transform.of(*trs[i]), oracle.WARP_BILINEAR_CV, border=1, max_value=1023) for i in range(n)])
    dp, dfs = 3 * w + 5, h * (3 * w + 5) + 3
    dst = np.full(offset + n * dfs + 8, 777, np.uint16)
    arr = (vs.Transform * n)(*[vs.Transform.of(*t) for t in trs])
    r = vs.lib().vs_bgr_image_warp_batch(C.c_void_p(frames.ctypes.data), h * w * 3, n, w, h, 3 * w, 3, 16, arr, vs.WARP_BILINEAR_CV, vs.BORDER_CONSTANT, 1023,
                                         C.c_void_p(dst.ctypes.data + 2 * offset), dfs, dp, vs.MEM_HOST, None)
    assert r >= 0, vs.lib().vs_last_error()
    keep = np.ones(dst.shape, bool)
    for i in range(n):
        rows = np.lib.stride_tricks.as_strided(dst[offset + i * dfs:], (h, 3 * w), (dp * 2, 2))
        assert same(rows, exp[i].reshape(h, 3 * w)), (i, offset)
        np.lib.stride_tricks.as_strided(keep[offset + i * dfs:], (h, 3 * w), (dp, 1))[...] = False
    assert np.all(dst[keep] == 777)
