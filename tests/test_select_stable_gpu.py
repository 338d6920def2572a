"""VS_SELECT_STABLE: the keep-best-fraction step under a documented, STL-independent rule -- smallest by
(abs_delta, tile index), survivors in ascending tile order -- against its oracle twin (select rule 1), bit for bit in everything
that is integer and to the usual 1e-12-grade agreement in the transforms (same survivors in the same order => the same sums)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def _d(tg, tc):
    return float(np.abs(np.array(tg.tup()) - np.array(tc.tup())).max())


def _check(res):
    for i, (ok_g, t_g, inf, ok_c, t_c, dbg) in enumerate(res):
        assert ok_g == ok_c and inf.fail_reason == dbg.fail_reason, (i, ok_g, ok_c, inf.fail_reason, dbg.fail_reason)
        if ok_c or dbg.fail_reason in (2, 3):
            assert list(inf.iterations[:dbg.levels]) == list(dbg.iterations[:dbg.levels]), i
            for l in range(dbg.levels):
                if dbg.iterations[l]:
                    assert abs(inf.condition[l] - dbg.condition[l]) <= 1e-9 * dbg.condition[l]
                assert (inf.selected_x[l], inf.selected_y[l]) == (dbg.selected_x[l], dbg.selected_y[l]), (i, l)
                assert _d(inf.level_transform[l], dbg.level_transform[l]) < TIGHT, (i, l)
        assert _d(t_g, t_c) < TIGHT, (i, t_g.tup(), t_c.tup())


@pytest.mark.parametrize("w,h,ch,kw", [(640, 480, 1, {}), (1920, 1080, 3, dict(pyramid_min_width=256)), (322, 246, 3, {}),
                                       (3840, 2160, 1, dict(pyramid_min_width=256))])
def test_stable_rule_one_frame_at_a_time(gpu_vs, oracle, w, h, ch, kw):
    """AlignNextFrame pattern (latency mode: helper workgroups, pipelined loop) in VS_SELECT_STABLE against the oracle's rule 1"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(w, h, 5, seed=71, channels=ch)
    gpu = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    cpu = oracle.Aligner(select_rule=oracle.SELECT_STABLE, **kw)
    res = []
    for f in frames:
        ok_g, t_g = gpu.align_next(f)
        ok_c, t_c = cpu.align_next(f)
        res.append((ok_g, t_g, gpu.info(0), ok_c, t_c, cpu.debug()))
    assert sum(r[3] for r in res) >= 3
    _check(res)


def test_stable_rule_is_a_different_member_of_the_family(gpu_vs, oracle):
    """the rule changes which tied tiles survive and the order of the sums: the transforms agree with the libstdc++-order modes to
    sub-pixel accuracy, not bit for bit -- and the three device builds of the stable mode agree with each other exactly"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(960, 540, 40, seed=72, channels=3)
    kw = dict(pyramid_min_width=100)
    ref = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_DEVICE, **kw)
    s_ref, t_ref = ref.align_batch(frames)
    outs = []
    for shared in (0, 1):
        a = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
        if shared:
            a.set_batch_mode(gpu_vs.BATCH_SHARED)               # >= 32 pairs: the small-footprint build (virtual threads)
        outs.append(a.align_batch(frames))
    seq = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    one = [seq.align_next(f) for f in frames]                   # latency mode
    assert outs[0][0] == outs[1][0] == [int(o[0]) for o in one] == s_ref
    for i in range(len(frames)):
        assert outs[0][1][i].tup() == outs[1][1][i].tup() == one[i][1].tup(), i
        assert _d(outs[0][1][i], t_ref[i]) < 0.05, i
    assert any(outs[0][1][i].tup() != t_ref[i].tup() for i in range(1, len(frames)))
    cpu = oracle.Aligner(select_rule=oracle.SELECT_STABLE, **kw)
    for i, f in enumerate(frames[:6]):
        ok_c, t_c = cpu.align_next(f)
        assert ok_c == bool(outs[0][0][i]) and _d(outs[0][1][i], t_c) < TIGHT, i


def test_stable_rule_4k_shared_build_selects_on_global_scratch(gpu_vs, oracle):
    """4K level 0 (20736 tiles per set) in the small-footprint build: the histogram passes and the placement run on the pair's
    global scratch; same results as the exclusive build, and as the oracle"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(3840, 2160, 34, seed=73, channels=1)
    kw = dict(pyramid_min_width=256)
    a = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    b = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    b.set_batch_mode(gpu_vs.BATCH_SHARED)
    sa, ta = a.align_batch(frames)
    sb, tb = b.align_batch(frames)
    assert sa == sb and sum(sa) >= 30
    assert [t.tup() for t in ta] == [t.tup() for t in tb]
    cpu = oracle.Aligner(select_rule=oracle.SELECT_STABLE, **kw)
    for i, f in enumerate(frames[:3]):
        ok_c, t_c = cpu.align_next(f)
        assert ok_c == bool(sa[i]) and _d(ta[i], t_c) < TIGHT, i


def test_stable_rule_failures_and_16bit_deltas(gpu_vs, oracle):
    """failure protocol under the stable rule, and sources whose abs_delta exceeds 255 (the general three-pass cut): 16-bit gray
    is not an input format, so large deltas come from a high-contrast 8-bit pair that does not align at the coarsest level"""
    from video_stabilizer_amd import synth
    path = [(0, 0, 0, 0), (0, 0, 60.0, -45.0), (0, 0, 0, 0), (0.0, 0.3, 0, 0), (0, 0, 1, 1)]
    frames, _ = synth.make_clip(320, 240, 5, seed=51, path=path, margin=160)
    for kw in ({}, dict(max_iters=2)):
        gpu = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
        cpu = oracle.Aligner(select_rule=oracle.SELECT_STABLE, **kw)
        res = []
        for f in frames:
            ok_g, t_g = gpu.align_next(f)
            ok_c, t_c = cpu.align_next(f)
            res.append((ok_g, t_g, gpu.info(0), ok_c, t_c, cpu.debug()))
        _check(res)


def test_stabilizer_in_stable_mode_matches_the_oracle_with_rule_1(gpu_vs, oracle):
    """VideoStabilizer frame by frame with the stable selection on both sides: same has-output pattern, same pixels"""
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 14, seed=77, channels=3)
    kw = dict(lag=3, smoother_memory=2, crop_pixels=8, warp_mode=gpu_vs.WARP_LANCZOS2)
    g = gpu_vs.Stabilizer(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    c = oracle.Stabilizer(select_rule=oracle.SELECT_STABLE, **kw)
    n_out = 0
    for f in frames:
        og, oc = g.process(f), c.process(f)
        assert (og is None) == (oc is None)
        if og is not None:
            assert np.array_equal(og, oc)
            n_out += 1
    assert n_out == len(frames) - 3


@pytest.mark.parametrize("ty,tx,hi,frac", [(27, 48, 12, 0.8), (27, 48, 3, 0.8), (54, 96, 40, 0.8), (9, 16, 65536, 0.8), (54, 96, 65536, 0.8),
                                           (54, 96, 1000, 0.5), (3, 3, 2, 1.0), (4, 4, 9, 0.01), (1, 1, 5, 0.8), (2, 97, 300, 0.8),
                                           (108, 240, 7, 0.8), (108, 240, 65536, 0.8), (7, 37, 256, 0.99)])
def test_stable_selection_kernel_against_the_oracle(gpu_vs, oracle, ty, tx, hi, frac):
    """vs_select_smallest_stable on arbitrary u16 tables: 8-bit deltas (one histogram pass), deltas up to 65535 (the byte-by-byte
    cut, which the engine's 8-bit luma never reaches), heavy ties, levels too small for the lane-sliced histogram copies, every
    element kept / none kept, several arrays per launch"""
    rng = np.random.default_rng(1000 * ty + tx + hi)
    wd = rng.integers(0, hi, size=(3, ty, tx)).astype(np.uint16)
    wd[1, :, : tx // 2] = wd[1, 0, 0]                       # half of array 1 is one value
    if hi > 256:
        wd[2] = (wd[2] & 0xff00) | 7                        # array 2: the low byte never decides
    got = gpu_vs.select_smallest_stable(wd, frac)
    for k in range(3):
        want = oracle.select_smallest_stable(wd[k], frac)
        assert np.array_equal(got[k], want), k


def test_environment_default_selects_the_stable_rule(gpu_vs):
    """VS_SELECT_MODE=2: handles created without a call to vs_aligner_set_select_mode (facade classes, harness programs) start in
    VS_SELECT_STABLE; checked in a child process (the variable is read once) against this process's explicit mode"""
    import json
    import os
    import subprocess
    import sys
    from video_stabilizer_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    frames, _ = synth.make_clip(320, 240, 4, seed=79, channels=3)
    want = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE).align_batch(frames)
    other = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_DEVICE).align_batch(frames)
    code = (
        "import sys, json, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import torch, numpy as np\n"
        "from video_stabilizer_amd import capi, synth\n"
        "frames, _ = synth.make_clip(320, 240, 4, seed=79, channels=3)\n"
        "a = capi.Aligner.__new__(capi.Aligner)\n"
        "a.params = capi.aligner_params()\n"
        "a.h = capi.lib().vs_aligner_create(C.byref(a.params), 0)\n"          # no set_select_mode: the environment decides
        "st, ts = a.align_batch(frames)\n"
        "print(json.dumps([list(map(int, st)), [t.tup() for t in ts]]))\n" % root)
    env = dict(os.environ, VS_SELECT_MODE="2")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    st, ts = json.loads(out.stdout.strip().splitlines()[-1])
    assert st == [int(x) for x in want[0]]
    assert [tuple(t) for t in ts] == [t.tup() for t in want[1]]
    assert [tuple(t) for t in ts] != [t.tup() for t in other[1]]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_stable_rule_random_sizes_and_fractions(gpu_vs, oracle, seed):
    """odd sizes, 3 ... 6 pyramid levels, smallest_fraction from 0.05 to 1.0 (every tile kept), gray and BGR: stable mode against
    the oracle's rule 1, one frame at a time and as one batch"""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(900 + seed)
    w, h = int(rng.integers(90, 700)), int(rng.integers(70, 500))
    ch = int(rng.choice([1, 3]))
    frac = float(rng.choice([0.05, 0.33, 0.8, 0.97, 1.0]))
    pmw = int(rng.choice([20, 40]))
    frames, _ = synth.make_clip(w, h, 5, seed=1000 + seed, channels=ch)
    kw = dict(smallest_fraction=frac, pyramid_min_width=pmw, pyramid_min_height=pmw)
    lv, ww, hh = 0, w, h
    while True:
        lv += 1; ww //= 2; hh //= 2
        if not (ww >= pmw and hh >= pmw):
            break
    if lv < 3:
        pytest.skip("%dx%d gives %d levels" % (w, h, lv))
    gpu = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    cpu = oracle.Aligner(select_rule=oracle.SELECT_STABLE, **kw)
    res = []
    for f in frames:
        ok_g, t_g = gpu.align_next(f)
        ok_c, t_c = cpu.align_next(f)
        res.append((ok_g, t_g, gpu.info(0), ok_c, t_c, cpu.debug()))
    _check(res)
    bat = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw)
    sb, tb = bat.align_batch(frames)
    assert [int(x) for x in sb] == [int(r[0]) for r in res]
    assert [t.tup() for t in tb] == [r[1].tup() for r in res]
