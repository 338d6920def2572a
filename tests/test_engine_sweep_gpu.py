"""Seeded random sweep of AlignNextFrame against the CPU oracle: frame size, channel count, bit depth, aligner parameters
(alignment.hpp:9-21 VideoAlignerParams), selection mode and the camera path's roughness all drawn at random.  Same gate as
test_engine_gpu.py (its _check_seq): success / failure decisions, failure reasons, iteration counts per level, selected-point
counts, condition numbers and the per-level and final transforms (1e-4)."""
import os

import numpy as np
import pytest

from _diff import same

from test_engine_gpu import TOL, _check_seq, _cmp_transform, _diverged, _run_both

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))     # a soak run draws this many times the cases (seeds continue upward)


def _draw(rng):
    w = int(rng.integers(97, 700))
    h = int(rng.integers(65, 460))
    # at least three pyramid levels (alignment.cpp:185-195: halve while both extents stay >= the minimum), else the call is an error
    min_w = int(rng.integers(8, max(9, w // 4)))
    min_h = int(rng.integers(8, max(9, h // 4)))
    params = dict(pyramid_min_width=min_w, pyramid_min_height=min_h,
                  smallest_fraction=float(rng.choice([0.5, 0.65, 0.8, 0.9, 1.0])),
                  max_iters=int(rng.choice([3, 8, 24, 64])),
                  threshold=float(rng.choice([0.01, 0.05, 0.2])),
                  max_displacement=float(rng.choice([4.0, 16.0, 64.0])),
                  phase_correlate=int(rng.integers(0, 2)))
    return w, h, int(rng.choice([1, 3])), int(rng.choice([8, 8, 10])), params, float(rng.choice([0.5, 2.0, 6.0]))


@pytest.mark.parametrize("seed", range(96 * _SCALE))
def test_random_configuration_matches_the_oracle(gpu_vs, oracle, seed):
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(9000 + seed)
    w, h, ch, bits, params, rough = _draw(rng)
    if params["phase_correlate"]:
        # (the FFT of pyramid level 2 takes power-of-two-free sizes too, but the oracle's DFT is O(n^2) per row: keep those frames small)
        w, h = min(w, 360), min(h, 260)
        params["pyramid_min_width"] = min(params["pyramid_min_width"], w // 4)
        params["pyramid_min_height"] = min(params["pyramid_min_height"], h // 4)
    frames, _ = synth.make_clip(w, h, 5, seed=700 + seed, channels=ch, bits=bits, jitter_t=rough)
    if bits == 10 and ch == 1:
        frames = (frames >> 2).astype(np.uint8)            # (gray frames are 8-bit only: VS_FMT_GRAY8)
    mode = int(rng.choice([gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE, gpu_vs.SELECT_STABLE]))     # (stable: against the oracle's rule 1)
    gpu, cpu, res = _run_both(gpu_vs, oracle, frames, select_mode=mode, **params)
    _check_seq(res)
    assert not res[0][0]                                    # first frame: no previous frame (alignment.cpp:231-234)


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_random_stabilizer_configuration_matches_the_oracle(gpu_vs, oracle, seed):
    """processFrame (stabilizer.cpp:9-117) with random VideoStabilizerParams (stabilizer.hpp:13-30) and both warp modes / borders:
    same has-output sequence, same measurement / accumulated transforms, frames within 1 LSB (the gate of test_stabilizer_matches_oracle)."""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(12000 + seed)
    w, h = int(rng.integers(160, 520)), int(rng.integers(120, 360))
    bits = int(rng.choice([8, 8, 10]))
    crop = int(rng.integers(0, min(w, h) // 5))
    kw = dict(lag=int(rng.integers(1, 7)), smoother_memory=int(rng.integers(0, 6)), crop_pixels=crop,
              enable_smoother=int(rng.integers(0, 2)), warp_mode=int(rng.choice([gpu_vs.WARP_LANCZOS2, gpu_vs.WARP_BILINEAR, gpu_vs.WARP_LANCZOS2_FAST, gpu_vs.WARP_LANCZOS2_SEP, gpu_vs.WARP_BILINEAR_CV])),
              warp_border=int(rng.integers(0, 2)), min_disp=float(rng.choice([0.5, 2.0])), max_disp=float(rng.choice([8.0, 40.0])))
    kw["lambda"] = float(rng.choice([0.5, 2.0, 8.0]))
    n = 14
    frames, _ = synth.make_clip(w, h, n, seed=300 + seed, channels=3, bits=bits, jitter_t=float(rng.choice([1.0, 4.0, 12.0])))
    if rng.random() < 0.33:                                 # the documented STL-independent selection rule on both sides
        g, c = gpu_vs.Stabilizer(device=0, select_mode=gpu_vs.SELECT_STABLE, **kw), oracle.Stabilizer(select_rule=oracle.SELECT_STABLE, **kw)
    else:
        g, c = gpu_vs.Stabilizer(device=0, **kw), oracle.Stabilizer(**kw)
    produced = 0
    for i, f in enumerate(frames):
        og, oc = g.process(f), c.process(f)
        assert (og is None) == (oc is None), i
        mg, ag, sg = g.state()
        mc, ac, sc = c.state()
        assert sg == sc, i
        assert _cmp_transform(mg, mc) < TOL and _cmp_transform(ag, ac) < 10 * TOL, i
        if oc is not None:
            produced += 1
            assert og.shape == oc.shape == (h - 2 * crop, w - 2 * crop, 3) and og.dtype == oc.dtype
            d = np.abs(og.astype(np.int32) - oc.astype(np.int32))
            assert d.max() <= 1 and (d != 0).mean() < 1e-2, (i, int(d.max()), float((d != 0).mean()))
    assert produced == n - kw["lag"]


@pytest.mark.parametrize("seed", range(48 * _SCALE))
def test_random_batch_forms_equal_frame_at_a_time(gpu_vs, seed):
    """vs_aligner_align_batch / _clips and vs_stabilizer_process_batch are DEFINED as n successive per-frame calls (include/vs_amd.h): on random
    configurations -- batch sizes from 2 to 150 pairs (latency mode with helper workgroups below 128 pairs, the throughput builds above), both
    solver builds, the three selection modes, host and device memory, batches cut at random points -- every status and every transform is the
    per-frame one bit for bit, and so is every output frame."""
    import torch
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(77000 + seed)
    w, h = int(rng.integers(130, 420)), int(rng.integers(100, 300))
    ch, bits = int(rng.choice([1, 3])), int(rng.choice([8, 8, 10]))
    if ch == 1:
        bits = 8
    n = int(rng.choice([3, 7, 20, 151]))
    kw = dict(pyramid_min_width=int(rng.integers(16, max(17, w // 4))), pyramid_min_height=int(rng.integers(12, max(13, h // 4))),
              smallest_fraction=float(rng.choice([0.5, 0.8, 1.0])), max_iters=int(rng.choice([4, 24, 64])))
    mode = int(rng.choice([gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE, gpu_vs.SELECT_STABLE]))
    base, _ = synth.make_clip(w, h, min(n, 12), seed=4000 + seed, channels=ch, bits=bits, jitter_t=float(rng.choice([1.0, 5.0])))
    frames = np.ascontiguousarray(base[np.arange(n) % len(base)])          # (a long batch repeats a short clip: content is not the point here)
    one = gpu_vs.Aligner(device=0, select_mode=mode, **kw)
    ref = [one.align_next(f) for f in frames]
    bat = gpu_vs.Aligner(device=0, select_mode=mode, **kw)
    bat.set_batch_mode(int(rng.choice([gpu_vs.BATCH_EXCLUSIVE, gpu_vs.BATCH_SHARED])))
    cut = int(rng.integers(1, n))                                           # two calls: the sequence carries across them
    if rng.random() < 0.5:
        st, ts = bat.align_batch(frames[:cut])
        s2, t2 = bat.align_batch(frames[cut:])
    else:
        fmt = gpu_vs.FMT_GRAY8 if ch == 1 else (gpu_vs.FMT_BGR8 if bits == 8 else gpu_vs.FMT_BGR10)
        dev = torch.from_numpy(frames.view(np.int16) if frames.dtype == np.uint16 else frames).to("cuda:0")
        esz = frames.dtype.itemsize
        st, ts = bat.align_batch_device(dev.data_ptr(), cut, w, h, fmt)
        s2, t2 = bat.align_batch_device(dev.data_ptr() + cut * h * w * ch * esz, n - cut, w, h, fmt)
    st, ts = st + s2, ts + t2
    for i in range(n):
        assert bool(st[i]) == ref[i][0], (i, n, cut, mode)
        if not ref[i][0] and max(abs(v) for v in ref[i][1].tup() + ts[i].tup()) > 2.0 ** 31:
            # a refused frame whose iteration DIVERGED beyond 2^31 pixels (|T| ~ 1e25 in the soak's case 40): the sampling positions saturate in the
            # float -> int conversion and the latency-mode and batch kernels need not pick the same border pixels.  Compared by its refusal only.
            # (A refused frame that stayed inside the integer range IS compared bit for bit.)
            continue
        assert ts[i].tup() == ref[i][1].tup(), (i, n, cut, mode)
    if ch == 3 and n <= 20:
        skw = dict(lag=int(rng.integers(1, 5)), smoother_memory=int(rng.integers(0, 4)), crop_pixels=int(rng.integers(0, 20)),
                   warp_mode=int(rng.integers(0, 5)), warp_border=int(rng.integers(0, 2)), **kw)
        seq = gpu_vs.Stabilizer(device=0, select_mode=mode, **skw)
        outs = [seq.process(f) for f in frames]
        sb = gpu_vs.Stabilizer(device=0, select_mode=mode, **skw)
        o1, h1 = sb.process_batch(frames[:cut])
        o2, h2 = sb.process_batch(frames[cut:])
        ob, hb = np.concatenate([o1, o2], 0), h1 + h2
        for i in range(n):
            assert bool(hb[i]) == (outs[i] is not None), i
            if outs[i] is not None:
                assert same(ob[i], outs[i]), (i, skw)


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_random_clip_batches_equal_fresh_handles(gpu_vs, seed):
    """vs_aligner_align_clips / vs_stabilizer_process_clips: every clip as if it went through its own fresh handle (include/vs_amd.h), for random clip
    counts and lengths -- the device-resident form cuts the clips into groups whose warps run under the next group's alignment, and one long clip
    into time chunks (at most four of each), so group and chunk boundaries fall in different places from case to case."""
    import torch
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(88000 + seed)
    w, h = int(rng.integers(140, 330)), int(rng.integers(110, 250))
    bits = int(rng.choice([8, 8, 10]))
    n_clips, fpc = int(rng.integers(1, 10)), int(rng.choice([3, 6, 13, 29, 61]))
    if n_clips == 1:
        fpc = int(rng.choice([100, 131, 200]))                    # one long clip: the time-chunk path (>= 96 frames)
    skw = dict(lag=int(rng.integers(1, 6)), smoother_memory=int(rng.integers(0, 4)), crop_pixels=int(rng.integers(0, 16)),
               warp_mode=int(rng.integers(0, 5)), warp_border=int(rng.integers(0, 2)),
               pyramid_min_width=int(rng.integers(16, w // 4)), pyramid_min_height=int(rng.integers(12, h // 4)))
    base = [synth.make_clip(w, h, min(fpc, 10), seed=6000 + 10 * seed + c, channels=3, bits=bits, jitter_t=float(rng.choice([1.0, 4.0])))[0]
            for c in range(n_clips)]
    clips = [np.ascontiguousarray(b[np.arange(fpc) % len(b)]) for b in base]
    allf = np.concatenate(clips)
    c = skw["crop_pixels"]
    want, want_has, want_t = [], [], []
    for clip in clips:
        s = gpu_vs.Stabilizer(device=0, **skw)
        a = gpu_vs.Aligner(device=0, **{k: skw[k] for k in ("pyramid_min_width", "pyramid_min_height")})
        for f in clip:
            o = s.process(f)
            want_has.append(0 if o is None else 1)
            want.append(np.zeros((h - 2 * c, w - 2 * c, 3), allf.dtype) if o is None else o)
            want_t.append(a.align_next(f))
    want = np.stack(want)
    fmt = gpu_vs.FMT_BGR8 if bits == 8 else gpu_vs.FMT_BGR10
    dev = torch.from_numpy(allf.view(np.int16) if bits > 8 else allf).to("cuda:0")
    out = torch.zeros((len(allf), h - 2 * c, w - 2 * c, 3), dtype=dev.dtype, device="cuda:0")
    s = gpu_vs.Stabilizer(device=0, **skw)
    if n_clips == 1:
        r, has = s.process_batch_device(dev.data_ptr(), fpc, w, h, fmt, out.data_ptr())
    else:
        r, has = s.process_clips_device(dev.data_ptr(), n_clips, fpc, w, h, fmt, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    got = got.view(np.uint16) if bits > 8 else got
    assert has == want_has and r == sum(want_has), (n_clips, fpc, skw)
    for i in range(len(allf)):
        if want_has[i]:
            assert same(got[i], want[i]), (i, n_clips, fpc, skw)
    al = gpu_vs.Aligner(device=0, **{k: skw[k] for k in ("pyramid_min_width", "pyramid_min_height")})
    st, ts = al.align_clips(len(allf), n_clips, mem_ptr=dev.data_ptr(), w=w, h=h, fmt=fmt)
    for i in range(len(allf)):
        assert bool(st[i]) == want_t[i][0], i
        if want_t[i][0] or max(abs(v) for v in want_t[i][1].tup() + ts[i].tup()) <= 2.0 ** 31:
            assert ts[i].tup() == want_t[i][1].tup(), i


@pytest.mark.parametrize("seed", range(40 * _SCALE))
def test_tiny_top_levels_device_selection_equals_host_selection(gpu_vs, seed):
    """Pyramids whose top level has fewer than 64 tiles per point set (a 20 x 13 image: 60 tiles), tie-heavy content (posterised frames: most
    abs_delta values are equal): the on-device replica of std::nth_element must leave what the host's std::nth_element leaves, bit for bit.
    (Regression: the register-level rounds' two 64-byte rank tables used to borrow the first 128 bytes of a posR array that holds 2 * n bytes;
    with n < 64 the x-set's tables ran into the y-set's while both waves were at work -- found by this file's sweep on the bounds-checked build.)"""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(99000 + seed)
    w, h = int(rng.integers(150, 420)), int(rng.integers(100, 280))
    lv_max = min(int(np.log2(w / 8)), int(np.log2(h / 6)))       # (a top level of at least 8 x 6: levels below 4 x 4 are refused)
    lv = int(rng.integers(max(2, lv_max - 1), lv_max + 1))
    mw, mh = max(4, (w >> lv) + 1), max(3, (h >> lv) + 1)          # the level after the lv-th is refused: lv + 1 levels
    kw = dict(pyramid_min_width=mw, pyramid_min_height=mh, smallest_fraction=float(rng.choice([0.5, 0.8, 0.95])), max_iters=int(rng.choice([3, 24])),
              phase_correlate=int(rng.integers(0, 2)))
    frames, _ = synth.make_clip(w, h, 5, seed=8000 + seed, channels=1, jitter_t=float(rng.choice([1.0, 6.0])))
    q = int(rng.choice([1, 32, 64]))
    frames = (frames // q * q).astype(np.uint8)
    host = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_STL_HOST, **kw)
    dev = gpu_vs.Aligner(device=0, select_mode=gpu_vs.SELECT_DEVICE, **kw)
    for i, f in enumerate(frames):
        (ok_h, t_h), (ok_d, t_d) = host.align_next(f), dev.align_next(f)
        ih, idv = host.info(0), dev.info(0)
        assert ok_h == ok_d and ih.fail_reason == idv.fail_reason, (i, kw)
        if max(abs(v) for v in t_h.tup() + t_d.tup()) > 2.0 ** 31:
            continue                                            # (diverged beyond the integer range: see test_random_batch_forms_equal_frame_at_a_time)
        assert list(ih.iterations[:ih.levels]) == list(idv.iterations[:ih.levels]), (i, kw)
        for l in range(ih.levels):
            assert (ih.selected_x[l], ih.selected_y[l]) == (idv.selected_x[l], idv.selected_y[l]), (i, l)
            # the same survivors in the same order give the same Hessian sums: the condition number is the fingerprint of a level's selection
            assert abs(ih.condition[l] - idv.condition[l]) <= 1e-9 * abs(ih.condition[l]) + (1e-15 * ih.condition[l]) * ih.condition[l], (i, l, kw, ih.condition[l], idv.condition[l])
        assert _cmp_transform(t_h, t_d) < TOL, (i, kw, t_h.tup(), t_d.tup())


@pytest.mark.parametrize("seed", range(60 * _SCALE))
def test_random_pitched_and_unaligned_frames_align_like_dense_ones(gpu_vs, seed):
    """AlignNextFrame / the stabilizer on frames as callers hold them -- rows longer than the image (every other row unaligned), frame strides with
    slack, device pointers that do not start on a dword, host or device memory: the ingest kernel's aligned 12-byte group loads must give way to
    its byte path exactly where they have to.  Same statuses and transforms as the dense copy, bit for bit; same output frames."""
    import ctypes as C
    import torch
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(66000 + seed)
    w, h = int(rng.integers(130, 400)), int(rng.integers(100, 260))
    ch, bits = int(rng.choice([1, 3])), int(rng.choice([8, 8, 10]))
    if ch == 1:
        bits = 8
    n = int(rng.integers(2, 7))
    dt, esz = (np.uint8, 1) if bits == 8 else (np.uint16, 2)
    frames, _ = synth.make_clip(w, h, n, seed=9000 + seed, channels=ch, bits=bits)
    kw = dict(pyramid_min_width=int(rng.integers(16, w // 4)), pyramid_min_height=int(rng.integers(12, h // 4)))
    fmt = gpu_vs.FMT_GRAY8 if ch == 1 else (gpu_vs.FMT_BGR8 if bits == 8 else gpu_vs.FMT_BGR10)
    ref = gpu_vs.Aligner(device=0, **kw)
    st_ref, ts_ref = ref.align_batch(frames, fmt=fmt)
    sp = ch * w + int(rng.integers(0, 8))
    sfs = h * sp + int(rng.integers(0, 5))
    so = int(rng.integers(0, 4))
    buf = rng.integers(0, 256, so + n * sfs + 8).astype(dt)
    for i in range(n):
        np.lib.stride_tricks.as_strided(buf[so + i * sfs:], (h, ch * w), (sp * esz, esz))[...] = frames[i].reshape(h, ch * w)
    out = (gpu_vs.Transform * n)()
    st = (C.c_int32 * n)()
    al = gpu_vs.Aligner(device=0, **kw)
    on_device = rng.random() < 0.5
    if on_device:
        dev = torch.from_numpy(buf.view(np.int16) if esz == 2 else buf).to("cuda:0")
        base, mem = dev.data_ptr() + so * esz, gpu_vs.MEM_DEVICE
    else:
        base, mem = buf.ctypes.data + so * esz, gpu_vs.MEM_HOST
    r = gpu_vs.lib().vs_aligner_align_batch(al.h, C.c_void_p(base), sfs, n, w, h, sp, fmt, mem, C.byref(al.params), out, st)
    assert r >= 0, gpu_vs.lib().vs_last_error()
    for i in range(n):
        assert bool(st[i]) == bool(st_ref[i]), (i, w, h, ch, bits, sp, sfs, so, on_device)
        if not st[i] and not all(abs(v) <= 2.0 ** 31 for v in out[i].tup() + ts_ref[i].tup()):
            continue                                            # (refused after diverging beyond the integer range: see test_random_batch_forms_equal_frame_at_a_time)
        assert same(out[i].tup(), ts_ref[i].tup(), equal_nan=True), (i, w, h, ch, bits, sp, sfs, so, on_device)
        assert not st[i] or all(np.isfinite(out[i].tup())), i          # (a refused frame's estimate may be NaN; an aligned frame's never)


@pytest.mark.parametrize("seed", range(48 * _SCALE))
def test_extreme_parameters_and_degenerate_frames_match_the_oracle(gpu_vs, oracle, seed):
    """the corners of VideoAlignerParams (alignment.hpp:9-21) and of the input: keep-fraction 0.001 ... 1, one or two iterations, thresholds of 0 and 1e9,
    displacement limits of 0 and 1e12; flat, saturated, two-level and pure-noise frames, frames that repeat.  Whatever the reference does with them
    (usually: refuse the frame) the product does too -- same statuses, reasons, counts, and estimates wherever an estimate exists."""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(123000 + seed)
    w, h = int(rng.integers(100, 360)), int(rng.integers(80, 240))
    kw = dict(pyramid_min_width=int(rng.integers(12, w // 4)), pyramid_min_height=int(rng.integers(10, h // 4)),
              smallest_fraction=float(rng.choice([0.001, 0.01, 0.3, 1.0])), max_iters=int(rng.choice([1, 2, 64])),     # (0 iterations / fractions outside (0, 1] are argument errors)
              threshold=float(rng.choice([0.0, 1e-12, 0.01, 1e9])), max_displacement=float(rng.choice([0.0, 1e-3, 10.0, 1e12])),
              phase_correlate=int(rng.integers(0, 2)))
    clip, _ = synth.make_clip(w, h, 6, seed=31000 + seed, channels=1, jitter_t=float(rng.choice([0.0, 2.0])))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        clip[:] = int(rng.integers(0, 256))                          # flat: every gradient 0, every Hessian singular
    elif kind == 1:
        clip = (clip > 128).astype(np.uint8) * 255                   # two levels
    elif kind == 2:
        clip = rng.integers(0, 256, clip.shape, dtype=np.uint8)      # noise: nothing to align
    elif kind == 3:
        clip[1::2] = clip[0::2]                                      # every frame twice: zero motion
    elif kind == 4:
        clip[3] = 255 - clip[3]                                      # one inverted frame in the middle
    mode = int(rng.choice([gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE, gpu_vs.SELECT_STABLE]))
    gpu, cpu, res = _run_both(gpu_vs, oracle, clip, select_mode=mode, **kw)
    _check_seq(res)
    for ok_g, t_g, *_ in res:
        assert not ok_g or all(np.isfinite(t_g.tup())), kw           # no frame is ever reported aligned with a non-finite transform


@pytest.mark.parametrize("seed", range(36 * _SCALE))
def test_stabilizer_on_degenerate_sequences_and_deep_formats_matches_the_oracle(gpu_vs, oracle, seed):
    """processFrame (stabilizer.cpp:9-117) where its failure branches live: clips with flat, noisy, inverted and repeated frames (alignment refused ->
    the accumulated transform is reset, stabilizer.cpp:25-33), saturated 10- / 12- / 16-bit content, extreme aspect ratios.  Same has-output pattern,
    same measurement / accumulated transforms, frames within 1 LSB."""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(135000 + seed)
    shape = int(rng.integers(0, 3))
    w, h = [(int(rng.integers(160, 420)), int(rng.integers(120, 300))), (int(rng.integers(700, 1400)), int(rng.integers(48, 80))),
            (int(rng.integers(64, 100)), int(rng.integers(400, 700)))][shape]
    bits = int(rng.choice([8, 10, 12, 16]))
    fmt = {8: gpu_vs.FMT_BGR8, 10: gpu_vs.FMT_BGR10, 12: gpu_vs.FMT_BGR12, 16: gpu_vs.FMT_BGR16_FULL}[bits]
    ofmt = {8: oracle.FMT_BGR8, 10: oracle.FMT_BGR10, 12: oracle.FMT_BGR12, 16: oracle.FMT_BGR16_FULL}[bits]
    n = 12
    clip, _ = synth.make_clip(w, h, n, seed=52000 + seed, channels=3, bits=min(bits, 10), jitter_t=float(rng.choice([1.0, 5.0])))
    if bits > 10:
        clip = (clip.astype(np.uint32) << (bits - 10)).clip(0, (1 << bits) - 1).astype(np.uint16)
    hi = (1 << bits) - 1
    for i in rng.choice(n, size=int(rng.integers(1, 5)), replace=False):
        kind = int(rng.integers(0, 5))
        if kind == 0: clip[i] = int(rng.integers(0, hi + 1))
        elif kind == 1: clip[i] = rng.integers(0, hi + 1, clip[i].shape).astype(clip.dtype)
        elif kind == 2: clip[i] = hi - clip[i]
        elif kind == 3 and i > 0: clip[i] = clip[i - 1]
        else: clip[i] = hi                                                  # saturated
    m = min(w, h)
    kw = dict(lag=int(rng.integers(1, 5)), smoother_memory=int(rng.integers(0, 4)), crop_pixels=int(rng.integers(0, max(1, m // 6))),
              warp_mode=int(rng.integers(0, 5)), warp_border=int(rng.integers(0, 2)),
              pyramid_min_width=max(8, w // int(rng.choice([4, 8, 16]))), pyramid_min_height=max(8, h // int(rng.choice([4, 8, 16]))))
    g, c = gpu_vs.Stabilizer(device=0, **kw), oracle.Stabilizer(**kw)
    for i, f in enumerate(clip):
        og, oc = g.process(f, fmt=fmt), c.process(f, fmt=ofmt)
        assert (og is None) == (oc is None), (i, kw)
        mg, ag, sg = g.state()
        mc, ac, sc = c.state()
        assert sg == sc, (i, kw)
        if _diverged(mc) or _diverged(mg):
            continue                                                        # (a refused frame's diverged estimate: see test_engine_gpu._check_seq)
        assert _cmp_transform(mg, mc) < TOL and _cmp_transform(ag, ac) < 10 * TOL, (i, kw, mg.tup(), mc.tup(), ag.tup(), ac.tup())
        if oc is not None:
            d = np.abs(og.astype(np.int64) - oc.astype(np.int64))
            assert og.shape == oc.shape and d.max() <= 1 and (d != 0).mean() < 1e-2, (i, kw, int(d.max()), float((d != 0).mean()))


@pytest.mark.parametrize("seed", range(30 * _SCALE))
def test_one_long_lived_handle_through_random_call_sequences(gpu_vs, seed):
    """One aligner and one stabilizer handle driven through a random sequence of calls -- frame at a time, batches of changing length (so the per-chunk
    storage regrows: allocate all -> copy the carry-over frame -> swap), frame sizes and formats that change (the reference restarts its sequence,
    alignment.cpp:357-367), explicit resets -- must answer every call the way a FRESH handle fed the same frames since the last restart does."""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(147000 + seed)
    kw = dict(pyramid_min_width=24, pyramid_min_height=18)
    skw = dict(lag=int(rng.integers(1, 4)), crop_pixels=int(rng.integers(0, 8)), warp_mode=int(rng.integers(0, 5)), **kw)
    mode = int(rng.integers(0, 3))
    A, S = gpu_vs.Aligner(device=0, select_mode=mode, **kw), gpu_vs.Stabilizer(device=0, select_mode=mode, **skw)
    fa = fs = None                                   # the fresh shadows, rebuilt at every restart
    size = None
    pos = 0
    for op in range(14):
        r = rng.random()
        if size is None or r < 0.2:                  # a new size / depth: both handles restart by themselves
            w, h, bits = int(rng.integers(100, 300)), int(rng.integers(80, 200)), int(rng.choice([8, 8, 10]))
            size = (w, h, bits)
            clip, _ = synth.make_clip(w, h, 40, seed=77000 + 20 * seed + op, channels=3, bits=bits, jitter_t=2.0)
            pos = 0
            fa, fs = gpu_vs.Aligner(device=0, select_mode=mode, **kw), gpu_vs.Stabilizer(device=0, select_mode=mode, **skw)
        elif r < 0.3:
            A.reset(); S.reset()
            fa, fs = gpu_vs.Aligner(device=0, select_mode=mode, **kw), gpu_vs.Stabilizer(device=0, select_mode=mode, **skw)
        k = int(rng.choice([1, 1, 2, 5, 11]))
        if pos + k > len(clip):
            pos = 0                                  # (the clip wraps: just more frames)
        fr = clip[pos:pos + k]
        pos += k
        if k == 1 and rng.random() < 0.6:
            got, want = A.align_next(fr[0]), fa.align_next(fr[0])
            assert got[0] == want[0] and same(got[1].tup(), want[1].tup(), equal_nan=True), (op, size)
            og, ow = S.process(fr[0]), fs.process(fr[0])
            assert (og is None) == (ow is None) and (og is None or same(og, ow)), (op, size)
        else:
            (st, ts), (st2, ts2) = A.align_batch(fr), fa.align_batch(fr)
            assert st == st2 and all(same(a.tup(), b.tup(), equal_nan=True) for a, b in zip(ts, ts2)), (op, size, k)
            (o, hs), (o2, hs2) = S.process_batch(fr), fs.process_batch(fr)
            assert hs == hs2 and same(o[np.array(hs, bool)], o2[np.array(hs2, bool)]), (op, size, k)


@pytest.mark.parametrize("seed", range(24 * _SCALE))
def test_tiny_top_levels_and_tie_heavy_frames_match_the_oracle_in_every_selection_mode(gpu_vs, oracle, seed):
    """the same corner as test_tiny_top_levels_device_selection_equals_host_selection (top levels below 64 tiles, posterised frames) against the
    ORACLE, in all three selection modes -- the stable rule's side-by-side histograms see the same tiny, tie-heavy tables"""
    from video_stabilizer_amd import synth
    rng = np.random.default_rng(171000 + seed)
    w, h = int(rng.integers(150, 420)), int(rng.integers(100, 280))
    lv_max = min(int(np.log2(w / 8)), int(np.log2(h / 6)))
    lv = int(rng.integers(max(2, lv_max - 1), lv_max + 1))
    kw = dict(pyramid_min_width=max(4, (w >> lv) + 1), pyramid_min_height=max(3, (h >> lv) + 1), smallest_fraction=float(rng.choice([0.5, 0.8, 0.95, 1.0])),
              max_iters=int(rng.choice([3, 24])))
    frames, _ = synth.make_clip(w, h, 4, seed=12000 + seed, channels=1, jitter_t=float(rng.choice([1.0, 6.0])))
    q = int(rng.choice([1, 32, 64]))
    frames = (frames // q * q).astype(np.uint8)
    for mode in (gpu_vs.SELECT_STL_HOST, gpu_vs.SELECT_DEVICE, gpu_vs.SELECT_STABLE):
        gpu, cpu, res = _run_both(gpu_vs, oracle, frames, select_mode=mode, **kw)
        _check_seq(res)
