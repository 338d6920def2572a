"""The debug build with bounds-checked LDS / scratch indexing ("bounds asserts in debug kernels"; GPU
AddressSanitizer is not available on this pool).  tools/build_variant.sh bounds compiles vs_engine / vs_warp / vs_phase / vs_capi
with -DVS_DEBUG_BOUNDS: the selection arrays (introselect_*, stable_select), the gather / exchange / staging indices of the fused
aligner kernel, the warp's tile fill and tap windows and the FFT lines are indexed through vsd::Span / VS_BOUNDS_CHECK
(vs_device.hpp).  A violation is recorded (site, index, limit, workgroup, thread) and redirected to element 0, never executed.

This module (a) proves the checker reports (a deliberate violation), and (b) runs the existing selection / stable-selection /
tiny-table / 4K-global-scratch / warp / phase / config tests AGAINST THAT BUILD in a pytest of its own: every result must still be
bit-identical to the oracle (the checks change no arithmetic) and after every test the bounds record must be clean
(tests/conftest.py::_bounds_record_stays_clean).  The regular library carries no checks: its kernels are instruction for
instruction what they were (compared with hipcc -S when the accessor was introduced).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "video_stabilizer_amd", "variants", "libvs_amd_bounds.so")


@pytest.fixture(scope="module")
def bounds_lib(gpu_vs):
    # (built on demand, and again whenever a source of the library is newer than it: a stale variant would test yesterday's kernels)
    csrc = os.path.join(ROOT, "video_stabilizer_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp", ".inc", ".cpp"))] + [os.path.join(ROOT, "include", "vs_amd.h")]
    if not os.path.exists(LIB) or max(os.path.getmtime(f) for f in srcs) > os.path.getmtime(LIB):
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), "bounds"])
    assert os.path.exists(LIB)
    return LIB


def _child(code, lib):
    env = dict(os.environ, VS_AMD_LIB=lib, VS_BOUNDS_BUILD="1")
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], env=env, capture_output=True, text=True, timeout=600)


def test_regular_library_has_no_bounds_build_entry_points(gpu_vs):
    assert gpu_vs.lib().vs_debug_bounds_check() == -3 and gpu_vs.lib().vs_debug_bounds_selftest() == -3      # VS_ERR_UNSUPPORTED
    assert b"VS_DEBUG_BOUNDS" in gpu_vs.lib().vs_last_error()


def test_the_checker_reports_a_deliberate_violation_and_does_not_execute_it(bounds_lib):
    out = _child("from video_stabilizer_amd import capi\n"
                 "L = capi.lib()\n"
                 "print('clean', capi.debug_bounds_check())\n"
                 "print('selftest', L.vs_debug_bounds_selftest())\n"
                 "print('after', capi.debug_bounds_check())\n"
                 "print('again', capi.debug_bounds_check())\n", bounds_lib)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "clean (0, '')" in out.stdout
    assert "selftest 202" in out.stdout                      # element 0 (100) stood in for element 11, + element 2 (102)
    assert "after (1, " in out.stdout and "site 900, index 11, limit 8, workgroup 0, thread 3" in out.stdout and "vs_capi.hip" in out.stdout
    assert "again (0, '')" in out.stdout                     # the record is cleared by the check


def test_selection_warp_phase_and_config_tests_pass_on_the_bounds_build_with_a_clean_record(bounds_lib):
    """the modules whose kernels carry the checked accessors: all of the selection / stable-selection / contracted-warp / phase tests and the seeded
    random sweeps (sizes from 1 x 1, degenerate transforms, random aligner / stabilizer parameters), and from the two slow modules (their time is the CPU oracle's) the cases that reach code nothing else reaches -- the 4K level 0 on
    global scratch, sixteen pairs with helper workgroups (exchange arrays), a 4K frame and the 10-bit stabilizer share at full size.
    (The WHOLE -m gpu suite has been run against this build once per round: profiles/r04_bounds_build.md.)"""
    # (every fresh device allocation of these runs starts filled with 0xA5: nothing compared against the oracle may depend on it)
    env = dict(os.environ, VS_AMD_LIB=bounds_lib, VS_BOUNDS_BUILD="1", VS_TEST_POISON_ALLOC="165", VS_TEST_HOOKS="1")
    # (round 5: the separable Lanczos2 tests replace the contracted ones -- the two forms share the tile, fill and store code, the sweeps below still
    # draw all three -- and the fixed-point bilinear kernels' tests join: byte / word tiles, coordinate tables, sites 211-216)
    runs = [(["tests/test_select_gpu.py", "tests/test_select_stable_gpu.py", "tests/test_warp_sep_gpu.py", "tests/test_warp_cv_gpu.py", "tests/test_phase_gpu.py",
              "tests/test_warp_sweep_gpu.py", "tests/test_engine_sweep_gpu.py"], "not 4k_frame"),   # (the kernel-chain sweep runs the un-instrumented stage kernels of vs_kernels.hip: not here)   # (the 4K frames' time is the CPU oracle's; the second group has a 4K frame)
            (["tests/test_latency_mode_gpu.py", "tests/test_configs_gpu.py"],
             "coresident_build_at_4k or sixteen_pairs or c3_4k_bgr_lanczos2_warp or c5_one_gpus_share")]
    for mods, expr in runs:
        cmd = [sys.executable, "-m", "pytest", *mods, "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + (["-k", expr] if expr else [])
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
        tail = out.stdout[-3000:] + out.stderr[-1500:]
        assert out.returncode == 0, tail
        assert " passed" in out.stdout and "failed" not in out.stdout, tail
