"""Phase-correlation initialisation on the GPU: vs_phase_correlate and the aligner's phase_correlate mode against the
oracle.  The transform specification is shared (oracle/vs_phase.cpp), so surfaces, shifts and responses are compared
bit for bit, and the aligner's results as in test_engine_gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w)).astype(np.uint8)


@pytest.mark.parametrize("shape", [(270, 480), (540, 960), (120, 160), (61, 97), (135, 240), (33, 60), (8, 8), (75, 45), (1, 17),
                                   (300, 4096)])
def test_surface_and_peak_bit_exact(gpu_vs, oracle, shape):
    h, w = shape
    big = _texture(h + 16, w + 16, seed=h * 7 + w)
    a = np.ascontiguousarray(big[8:8 + h, 8:8 + w])
    b = np.ascontiguousarray(big[5:5 + h, 10:10 + w])       # content moved by (-2, +3)
    dx, dy, r, surf = gpu_vs.phase_correlate(a, b, want_surface=True)
    want = oracle.phase_surface(a.astype(np.float32), b.astype(np.float32))
    assert surf.shape == want.shape == (oracle.optimal_dft_size(h), oracle.optimal_dft_size(w))
    assert np.array_equal(surf, want)
    assert (dx, dy, r) == oracle.phase_correlate(a, b)
    assert gpu_vs.phase_correlate(a, b) == (dx, dy, r)      # without the surface output


def test_circular_shift_and_identical_images(gpu_vs, oracle):
    a = _texture(270, 480, seed=3)
    for d in [(0, 0), (3, -2), (-7, 5), (100, 60)]:
        b = np.roll(a, (d[1], d[0]), (0, 1))
        got = gpu_vs.phase_correlate(a, b)
        assert got == oracle.phase_correlate(a, b)
        assert abs(got[0] - d[0]) < 1e-5 and abs(got[1] - d[1]) < 1e-5 and abs(got[2] - 1) < 1e-5
    # flat images: every cross-power term is 0 / (0 + eps) except DC -> defined, equal on both sides
    z = np.full((64, 80), 7, np.uint8)
    assert gpu_vs.phase_correlate(z, z) == oracle.phase_correlate(z, z)
    k = np.zeros((64, 80), np.uint8)
    assert gpu_vs.phase_correlate(k, k) == oracle.phase_correlate(k, k)


def test_too_large_is_an_error(gpu_vs):
    a = np.zeros((4, 4100), np.uint8)
    with pytest.raises(gpu_vs.VsError, match="padded extent"):
        gpu_vs.phase_correlate(a, a)


def _cmp(tg, tc):
    return float(np.abs(np.array(tg.tup()) - np.array(tc.tup())).max())


@pytest.mark.parametrize("w,h,ch,kw", [(640, 480, 1, {}), (322, 246, 3, {}), (1920, 1080, 3, dict(pyramid_min_width=256)),
                                       (640, 480, 3, dict(phase_correlate_threshold=0.97))])
@pytest.mark.parametrize("select_mode", [0, 1])
def test_aligner_phase_mode_matches_oracle(gpu_vs, oracle, w, h, ch, kw, select_mode):
    from video_stabilizer_amd import synth
    n = 7
    frames, _ = synth.make_clip(w, h, n, seed=31 + w, channels=ch)
    gpu = gpu_vs.Aligner(device=0, select_mode=select_mode, phase_correlate=1, **kw)
    cpu = oracle.Aligner(phase_correlate=1, **kw)
    used = 0
    for i, f in enumerate(frames):
        ok_g, t_g = gpu.align_next(f)
        ok_c, t_c = cpu.align_next(f)
        inf, dbg = gpu.info(0), cpu.debug()
        assert ok_g == ok_c and inf.fail_reason == dbg.fail_reason, i
        assert (inf.phase_dx, inf.phase_dy, inf.phase_response) == (dbg.phase_dx, dbg.phase_dy, dbg.phase_response), i
        assert list(inf.iterations[:dbg.levels]) == list(dbg.iterations[:dbg.levels]), i
        assert _cmp(t_g, t_c) < 1e-9, (i, t_g.tup(), t_c.tup())
        if i:
            used += dbg.phase_response > kw.get("phase_correlate_threshold", 0.5)
    assert used > 0                       # the start value was actually applied on some pair
    if "phase_correlate_threshold" in kw:
        assert used < n - 1               # and rejected on others: both branches ran


def test_batch_equals_sequential_and_mode_can_change_per_call(gpu_vs, oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(640, 480, 9, seed=8, channels=3)
    seq = gpu_vs.Aligner(device=0, phase_correlate=1)
    want = [seq.align_next(f) for f in frames]
    bat = gpu_vs.Aligner(device=0, phase_correlate=1)
    st, ts = bat.align_batch(frames[:4])
    st2, ts2 = bat.align_batch(frames[4:])        # the carry-over frame's spectrum is rebuilt from its pyramid
    for i, (ok, t) in enumerate(want):
        assert bool((st + st2)[i]) == ok
        assert (ts + ts2)[i].tup() == t.tup(), i
    # clips: every clip is a fresh sequence
    clips = gpu_vs.Aligner(device=0, phase_correlate=1)
    stc, tsc = clips.align_clips(np.concatenate([frames[:3], frames[:3]]), 2)
    assert [t.tup() for t in tsc[:3]] == [t.tup() for t in tsc[3:]] == [t.tup() for _, t in want[:3]]
    # phase_correlate switched on for the second call only: the reference builds PhaseImage for every frame
    # (alignment.cpp:225-229), so the pair (frame 3, frame 4) is correlated
    mixed = gpu_vs.Aligner(device=0)
    cpu = oracle.Aligner()
    for f in frames[:4]:
        mixed.align_next(f)
        cpu.align_next(f)
    p_on = gpu_vs.aligner_params(phase_correlate=1)
    mixed.params = p_on
    cpu.params = oracle.aligner_params(phase_correlate=1)
    ok_g, t_g = mixed.align_next(frames[4])
    ok_c, t_c = cpu.align_next(frames[4])
    assert ok_g == ok_c and _cmp(t_g, t_c) < 1e-9
    assert mixed.info(0).phase_response == cpu.debug().phase_response > 0


def test_stabilizer_with_phase_correlation(gpu_vs, oracle):
    from video_stabilizer_amd import synth
    frames, _ = synth.make_clip(320, 240, 16, seed=5, channels=3)
    g = gpu_vs.Stabilizer(device=0, lag=3, phase_correlate=1)
    c = oracle.Stabilizer(lag=3, phase_correlate=1)
    outs = 0
    for f in frames:
        og, oc = g.process(f), c.process(f)
        assert (og is None) == (oc is None)
        if og is not None:
            outs += 1
            assert np.array_equal(og, oc)
    assert outs == len(frames) - 3
    # the stage timer sees the mode
    a = gpu_vs.Aligner(device=0, phase_correlate=1)
    a.enable_timing(True)
    a.align_batch(frames)
    tm = a.timings()
    assert tm["phase"]["launches"] == 2 + 4 and tm["phase"]["ms"] > 0
