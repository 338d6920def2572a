"""Latency mode of the fused aligner kernel (few pairs in flight): the pipelined one-barrier Gauss-Newton loop on the
LDS-resident level and the helper workgroups that share the sparse_warpdiff pass of the large levels must not change a
bit of the result.  The switches are read once per process (VS_GN_PIPELINE, VS_GN_HELPERS), so every configuration runs in
a process of its own; the frames come from the same seeded generator.

  plain    VS_GN_PIPELINE=0 VS_GN_HELPERS=1   one workgroup per pair, two-barrier loop (what a full batch runs)
  default  (unset)                             pipelined loop + 16 workgroups per pair for <= 16 pairs
  few      VS_GN_HELPERS=2 / 3                 few, large slices: several staging passes per helper
  sync     VS_GN_POLL=0                        completion through hipStreamSynchronize instead of the polled result block
  cores    VS_GN_CORESIDENT=1 / 0              the small-footprint build of the fused kernel (256 hardware threads standing in for
                                               the 512 of the plain build: virtual threads) / the plain build for a full batch
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import torch                      # before the library: one HIP runtime per process (tests/conftest.py)
torch.cuda.init()
from video_stabilizer_amd import capi, synth
w, h, n, batch = (int(a) for a in sys.argv[2:6])
# (the numpy generator takes ~1.5 s per 4K frame and every configuration is a process of its own: the clip is generated once per size and
# handed from child to child through a file -- same seeded frames, a prefix of the longest clip made so far)
import os, tempfile, numpy as np
cache = os.path.join(tempfile.gettempdir(), "vs_latency_clip_%dx%d_seed77_uid%d.npy" % (w, h, os.getuid()))
frames = None
try:
    c = np.load(cache, mmap_mode="r")
    if c.shape[0] >= n and c.shape[1:] == (h, w, 3): frames = np.ascontiguousarray(c[:n])
except Exception:
    frames = None
if frames is None:
    frames, _ = synth.make_clip(w, h, n, seed=77, channels=3)
    try:
        tmp = cache + ".%d.tmp.npy" % os.getpid()
        np.save(tmp, frames)
        os.replace(tmp, cache)
    except Exception:
        pass
al = capi.Aligner(device=0, pyramid_min_width=int(sys.argv[6]) if len(sys.argv) > 6 else 256)
if len(sys.argv) > 7 and int(sys.argv[7]): al.set_batch_mode(capi.BATCH_SHARED)
out = []
def rec(i, st, t):
    inf = al.info(i)
    out.append([int(st), [float(x).hex() for x in t.tup()], int(inf.fail_reason), [int(x) for x in inf.iterations[:inf.levels]],
                [float(c).hex() for c in inf.condition[:inf.levels]],
                [[float(x).hex() for x in inf.level_transform[l].tup()] for l in range(inf.levels)], list(inf.selected_x[:inf.levels])])
if batch:
    st, ts = al.align_batch(frames)
    for i in range(n): rec(i, st[i], ts[i])
else:
    for i in range(n):
        ok, t = al.align_next(frames[i])
        rec(0, ok, t)
print("RESULT " + json.dumps(out))
"""


def _run(w, h, n, batch, pmw=256, shared=0, **env):
    e = dict(os.environ)
    for k in ("VS_GN_PIPELINE", "VS_GN_HELPERS", "VS_GN_STALL_HELPERS", "VS_GN_POLL", "VS_GN_CORESIDENT", "VS_GN_SELECT_DEPTH"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(w), str(h), str(n), str(int(batch)), str(pmw), str(int(shared))], env=e, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_1080p_one_frame_at_a_time_equals_the_plain_launch():
    plain = _run(1920, 1080, 5, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    assert sum(r[0] for r in plain) == 4 and all(r[2] == 0 for r in plain[1:])
    assert _run(1920, 1080, 5, False) == plain                                  # pipelined + 16 workgroups per pair
    assert _run(1920, 1080, 5, False, VS_GN_PIPELINE=1, VS_GN_HELPERS=1) == plain
    assert _run(1920, 1080, 5, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=3) == plain


def test_1080p_small_batch_equals_the_plain_launch():
    # four pairs in one launch: 64 workgroups, every leader with its own helpers
    plain = _run(1920, 1080, 5, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    assert _run(1920, 1080, 5, True) == plain


def test_sixteen_pairs_in_one_launch_equal_the_plain_launch():
    # the largest launch that still gets helpers (16 pairs x 16 workgroups) and the direct (kernel-argument) descriptors
    plain = _run(1920, 1080, 17, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    assert sum(r[0] for r in plain) == 16
    assert _run(1920, 1080, 17, True) == plain


def test_mid_size_launch_shares_the_idle_cus():
    # 17 .. 128 pairs: as many workgroups per pair as keep the launch within one per CU (40 pairs -> 6 per pair), descriptors
    # and states through device memory (the direct form stops at 16 pairs)
    plain = _run(1920, 1080, 41, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    assert sum(r[0] for r in plain) == 40
    assert _run(1920, 1080, 41, True) == plain


def test_polled_and_synchronised_completion_agree():
    assert _run(1920, 1080, 4, False, VS_GN_POLL=0) == _run(1920, 1080, 4, False)


def test_4k_set_by_set_levels_equal_the_plain_launch():
    # 4K level 0 has 20736 tiles per set: selected set by set, keys and samples read back in two halves; with two
    # workgroups per pair a slice does not fit the staging area in one pass
    plain = _run(3840, 2160, 3, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    assert sum(r[0] for r in plain) == 2
    assert _run(3840, 2160, 3, False) == plain
    assert _run(3840, 2160, 3, False, VS_GN_HELPERS=2) == plain


def test_a_helper_that_never_reports_back_costs_a_timeout_not_a_hang():
    # every wait inside the kernel is bounded: with the helpers silenced (test hook) the leader gives up after ~0.3 s per call,
    # flags the pair, and the engine redoes it through the per-level host path -- same result as the plain launch
    plain = _run(1920, 1080, 3, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1)
    stalled = _run(1920, 1080, 3, False, VS_GN_STALL_HELPERS=1)
    assert stalled == plain


def test_coresident_build_equals_the_plain_launch_bit_for_bit():
    # the 256-thread build walks the virtual threads / waves of the 512-thread build: same additions, same order, same bits --
    # transforms, iteration counts, condition numbers, failure codes
    plain = _run(1920, 1080, 5, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    assert sum(r[0] for r in plain) == 4
    assert _run(1920, 1080, 5, True, VS_GN_CORESIDENT=1) == plain
    # one frame at a time through the small build too (direct descriptors, polled result block)
    assert _run(1920, 1080, 5, False, VS_GN_CORESIDENT=1) == _run(1920, 1080, 5, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)


def test_shared_mode_full_batch_equals_the_plain_launch():
    # vs_aligner_set_batch_mode(VS_BATCH_SHARED), 32 pairs or more: the small-footprint build; 640x480 with the reference's default pyramid keeps the
    # clip small (5 levels of 1200 / 300 tiles per set)
    plain = _run(640, 480, 140, True, pmw=20, VS_GN_CORESIDENT=0)
    assert sum(r[0] for r in plain) >= 130
    assert _run(640, 480, 140, True, pmw=20, shared=1) == plain
    assert _run(640, 480, 140, True, pmw=20, VS_GN_CORESIDENT=1) == plain


def test_coresident_build_on_odd_sizes_and_default_pyramid():
    # 1280x720 with pyramid_min_width 256: 3 levels, tile sizes 20 / 14 / 6 with remainder pixels, 2304 / 1125 / 1590 tiles per
    # set -- every level selects both sets side by side, 128 hardware threads each
    plain = _run(1280, 720, 6, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    assert _run(1280, 720, 6, True, VS_GN_CORESIDENT=1) == plain


def test_depth_limit_exit_is_redone_on_the_host_with_the_same_result():
    # fail_reason 100 ("libstdc++ would have heap-selected": tests/test_select_gpu.py shows real tables that do it) sends the
    # whole chunk through the per-level path with the host's std::nth_element.  With the on-device budget cut to 3 rounds (test
    # hook) every ordinary pair takes that exit; transforms, iteration counts, condition numbers, per-level transforms must
    # equal the plain run -- one pair at a time, a small batch, a full batch (both kernel builds)
    plain = _run(1920, 1080, 5, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    assert _run(1920, 1080, 5, True, VS_GN_SELECT_DEPTH=3) == plain
    assert _run(1920, 1080, 5, False, VS_GN_SELECT_DEPTH=3) == _run(1920, 1080, 5, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    assert _run(1920, 1080, 5, True, VS_GN_SELECT_DEPTH=3, VS_GN_CORESIDENT=1) == plain
    big = _run(640, 480, 140, True, pmw=20, VS_GN_CORESIDENT=0)
    assert _run(640, 480, 140, True, pmw=20, VS_GN_SELECT_DEPTH=2) == big


def test_coresident_build_at_4k_selects_the_finest_level_on_global_scratch():
    # a 4K level 0 has 20736 tiles per point set: 124 KB of selection arrays cannot live in the small build's 32 KB of LDS, so that
    # level's introselect runs on the pair's global scratch (128 elements per thread, the tail in LDS) -- same permutation, same bits
    plain = _run(3840, 2160, 4, True, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    assert sum(r[0] for r in plain) == 3
    assert _run(3840, 2160, 4, True, VS_GN_CORESIDENT=1) == plain
    assert _run(3840, 2160, 4, False, VS_GN_CORESIDENT=1) == _run(3840, 2160, 4, False, VS_GN_PIPELINE=0, VS_GN_HELPERS=1, VS_GN_CORESIDENT=0)
    # the depth-limit exit of the global-scratch selection goes through the same host redo
    assert _run(3840, 2160, 3, True, VS_GN_CORESIDENT=1, VS_GN_SELECT_DEPTH=3) == plain[:3]
