"""Every BASELINE.json config at its real frame size against the CPU oracle (configs[2..4]; configs[0..1] are in
test_engine_gpu.py).  Clips are made on the device by synth.TorchClipFactory (the generator bench.py uses) and copied
to the host once, so the GPU path and the oracle see identical bytes.

  C2 / C3  the WHOLE headline clips (240 x 1080p, 120 x 4K: bench.py's generator and seeds) through the aligner == oracle, frame for frame
  C3  4K BGR: bgr_image_warp Lanczos2 of a whole frame == oracle, bit for bit; 4K BGR alignment (4 levels) == oracle
  C4  many independent 1080p clips through vs_aligner_align_clips == one oracle aligner per clip
  C5  10-bit BGR through the full stabilizer loop (vs_stabilizer_process_clips) == one oracle stabilizer per clip;
      one 3840x2160 10-bit clip of lag + 2 frames through size-independent properties; one 3840x2160 10-bit clip through the loop against the
      oracle stabilizer frame for frame (the default fixed-point bilinear warp and the separable Lanczos2)
  C4 / C5 at full size: one GPU's share of the 64-clip configs (8 clips x 120 x 1080p; 8 clips x 60 x 4K 10-bit), resident
      in HBM, in one call -- clip independence (a clip re-run alone is bit-identical), latency pattern, value range
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _device_clip(w, h, n, seed, bits=8, **path_kw):
    import torch
    from video_stabilizer_amd import synth
    t, path = synth.make_clip_torch(w, h, n, seed, torch.device("cuda", 0), channels=3, bits=bits, **path_kw)
    a = t.cpu().numpy()
    return (a.view(np.uint16) if bits != 8 else a), path


def _same_alignment(inf, dbg, ok_g, ok_c, t_g, t_c, tag):
    assert bool(ok_g) == bool(ok_c), (tag, ok_g, ok_c, inf.fail_reason, dbg.fail_reason)
    assert inf.fail_reason == dbg.fail_reason, tag
    assert float(np.abs(np.array(t_g.tup()) - np.array(t_c.tup())).max()) < TOL, (tag, t_g.tup(), t_c.tup())
    if ok_c:
        assert list(inf.iterations[:dbg.levels]) == list(dbg.iterations[:dbg.levels]), tag


def test_c3_4k_bgr_lanczos2_warp_is_the_oracle_bit_for_bit(gpu_vs, oracle):
    frames, _ = _device_clip(3840, 2160, 1, seed=2)
    for tr in [(0.002, -0.0015, 3.3, -2.7), (-0.004, 0.006, -11.5, 7.25)]:
        got = gpu_vs.bgr_image_warp(frames[0], gpu_vs.Transform.of(*tr))
        want = oracle.bgr_image_warp(frames[0], oracle.Transform.of(*tr))
        assert np.array_equal(got, want), tr


def test_c3_4k_bgr_alignment_matches_the_oracle(gpu_vs, oracle):
    frames, _ = _device_clip(3840, 2160, 4, seed=2)
    kw = dict(pyramid_min_width=256)                        # 4 levels at 4K
    gpu, cpu = gpu_vs.Aligner(device=0, **kw), oracle.Aligner(**kw)
    st, ts = gpu.align_batch(frames)
    good = 0
    for i, f in enumerate(frames):
        ok_c, t_c = cpu.align_next(f)
        dbg = cpu.debug()
        assert dbg.levels == 4 or i == 0
        _same_alignment(gpu.info(i), dbg, st[i], ok_c, ts[i], t_c, i)
        good += bool(ok_c)
    assert good == 3


@pytest.mark.parametrize("w,h,n,seed,levels", [(1920, 1080, 240, 1, 3), (3840, 2160, 120, 2, 4)])
def test_whole_headline_clips_match_the_oracle_aligner(gpu_vs, oracle, w, h, n, seed, levels):
    """BASELINE configs[1] and configs[2] at their real size AND length: the clip bench.py times (same generator, same seed) -- 240 frames of 1080p,
    120 frames of 4K -- through vs_aligner_align_batch on device memory in one call, against the oracle aligner frame for frame: status, failure
    reason, iterations per level, transform within 1e-4."""
    import torch
    from video_stabilizer_amd import synth
    dev = torch.device("cuda", 0)
    clip = synth.TorchClipFactory(w, h, seed, dev, channels=3, bits=8).make(n, seed)[0]
    torch.cuda.synchronize()
    kw = dict(pyramid_min_width=256)
    gpu, cpu = gpu_vs.Aligner(device=0, **kw), oracle.Aligner(**kw)
    st, ts = gpu.align_batch_device(clip.data_ptr(), n, w, h, gpu_vs.FMT_BGR8)
    host = clip.cpu().numpy()
    good = 0
    for i in range(n):
        ok_c, t_c = cpu.align_next(host[i])
        dbg = cpu.debug()
        assert i == 0 or dbg.levels == levels
        _same_alignment(gpu.info(i), dbg, st[i], ok_c, ts[i], t_c, i)
        good += bool(ok_c)
    assert good == n - 1                                   # what bench.py counts as `aligned_per_step`


def test_c4_many_1080p_clips_match_one_oracle_aligner_per_clip(gpu_vs, oracle):
    n_clips, fpc = 4, 6
    kw = dict(pyramid_min_width=256)                        # 3 levels at 1080p
    clips = [_device_clip(1920, 1080, fpc, seed=1000 + c)[0] for c in range(n_clips)]
    gpu = gpu_vs.Aligner(device=0, **kw)
    st, ts = gpu.align_clips(np.concatenate(clips, 0), n_clips)
    good = 0
    for c in range(n_clips):
        cpu = oracle.Aligner(**kw)                          # a fresh VideoAligner per clip, as grid_search_align.cpp:174 does
        for k in range(fpc):
            ok_c, t_c = cpu.align_next(clips[c][k])
            i = c * fpc + k
            _same_alignment(gpu.info(i), cpu.debug(), st[i], ok_c, ts[i], t_c, (c, k))
            good += bool(ok_c)
        assert st[c * fpc] == 0 and gpu.info(c * fpc).fail_reason == 1     # every clip starts with a first frame
    assert good == n_clips * (fpc - 1)


@pytest.mark.parametrize("sampler", ["default", "lanczos2"])
def test_c5_10bit_stabilizer_clips_match_one_oracle_stabilizer_per_clip(gpu_vs, oracle, sampler):
    n_clips, fpc, w, h = 2, 14, 640, 360
    clips = [_device_clip(w, h, fpc, seed=2000 + c, bits=10)[0] for c in range(n_clips)]
    # VideoStabilizerParams defaults: lag 10, smoother 5, crop 32, and the reference's bilinear warp; "lanczos2" = bgr_image_warp
    kw = {} if sampler == "default" else dict(warp_mode=gpu_vs.WARP_LANCZOS2)
    g = gpu_vs.Stabilizer(device=0, **kw)
    assert g.params.warp_mode == (gpu_vs.WARP_BILINEAR_CV if sampler == "default" else gpu_vs.WARP_LANCZOS2)
    out, has = g.process_clips(np.concatenate(clips, 0), n_clips)
    produced = 0
    for c in range(n_clips):
        cpu = oracle.Stabilizer(**kw)
        for k in range(fpc):
            oc = cpu.process(clips[c][k])
            i = c * fpc + k
            assert bool(has[i]) == (oc is not None), (c, k)
            if oc is None:
                continue
            produced += 1
            og = out[i]
            assert og.shape == oc.shape == (h - 64, w - 64, 3) and og.dtype == np.uint16
            assert int(og.max()) <= 1023                    # 10-bit content stays 10-bit (Lanczos overshoot is clamped)
            d = np.abs(og.astype(np.int32) - oc.astype(np.int32))
            # the two warps sample with transforms that agree to 1e-4 px: a value may land on the other side of a
            # rounding boundary, never further
            assert d.max() <= 1 and (d != 0).mean() < 1e-2, (c, k, int(d.max()), float((d != 0).mean()))
    assert produced == n_clips * (fpc - 10)


@pytest.mark.parametrize("sampler,fpc,lag", [("default", 60, 10), ("separable", 14, 10)])
def test_c5_full_size_clip_matches_the_oracle_stabilizer(gpu_vs, oracle, sampler, fpc, lag):
    """configs[4] at its REAL frame size against the oracle ("C5 at full size is property-checked only"): one 3840 x 2160 10-bit
    clip through the full stabilizer loop -- the library defaults (lag 10, L1 smoother, crop 32, cv::warpAffine's fixed-point bilinear on 16-bit
    containers: a WHOLE 60-frame clip of the config, 50 outputs) and the separable Lanczos2 (14 frames, 4 outputs: the oracle's Lanczos2 takes seconds per 4K frame) --
    frame for frame against one oracle stabilizer: same latency pattern, pixels within the 1-LSB band two warps leave whose transforms agree to 1e-4 px."""
    w, h = 3840, 2160
    clip = _device_clip(w, h, fpc, seed=2005, bits=10)[0]
    kw = dict(pyramid_min_width=256)
    if lag != 10:
        kw.update(lag=lag, smoother_memory=2)
    if sampler != "default":
        kw.update(warp_mode=gpu_vs.WARP_LANCZOS2_SEP, warp_border=gpu_vs.BORDER_CLAMP)
    g = gpu_vs.Stabilizer(device=0, **kw)
    out, has = g.process_batch(clip)
    okw = dict(kw)
    if sampler != "default":
        okw.update(warp_mode=oracle.WARP_LANCZOS2_SEPARABLE, warp_border=oracle.BORDER_CLAMP)
    cpu = oracle.Stabilizer(**okw)
    produced = 0
    for k in range(fpc):
        oc = cpu.process(clip[k])
        assert bool(has[k]) == (oc is not None) == (k >= lag), k
        if oc is None:
            continue
        produced += 1
        og = out[k]
        assert og.shape == oc.shape == (h - 64, w - 64, 3) and og.dtype == np.uint16 and int(og.max()) <= 1023
        d = np.abs(og.astype(np.int32) - oc.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-2, (sampler, k, int(d.max()), float((d != 0).mean()))
    assert produced == fpc - lag


def test_c5_4k_10bit_clip_properties(gpu_vs):
    """3840x2160 10-bit through the full loop at the config's real size (the oracle takes minutes there): the properties
    that do not depend on size -- first `lag` calls give nothing, clips == sequential calls bit for bit, 10-bit range kept,
    and a static clip comes back as the crop of its input within 1 LSB (identity correction)."""
    lag, w, h = 10, 3840, 2160
    frames, _ = _device_clip(w, h, lag + 2, seed=2000, bits=10)
    s = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_LANCZOS2)
    out, has = s.process_clips(frames, 1)
    assert has == [0] * lag + [1, 1] and out.shape == (lag + 2, h - 64, w - 64, 3)
    assert int(out[lag:].max()) <= 1023
    seq = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_LANCZOS2)
    for i, f in enumerate(frames):
        o = seq.process(f)
        assert (o is None) == (i < lag)
        if o is not None:
            assert np.array_equal(o, out[i]), i
    static = np.repeat(frames[:1], lag + 2, axis=0)
    o2, h2 = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_LANCZOS2).process_clips(static, 1)
    assert h2 == has
    d = np.abs(o2[lag].astype(np.int32) - frames[0][32:-32, 32:-32].astype(np.int32))
    assert d.max() <= 1


def _device_clips(w, h, n_clips, fpc, seed, bits=8):
    """n_clips clips back to back in ONE device tensor (the layout vs_aligner_align_clips / vs_stabilizer_process_clips take)"""
    import torch
    from video_stabilizer_amd import synth
    dev = torch.device("cuda", 0)
    fac = synth.TorchClipFactory(w, h, seed, dev, channels=3, bits=bits)
    allf = torch.empty((n_clips * fpc, h, w, 3), dtype=torch.uint8 if bits == 8 else torch.int16, device=dev)
    for c in range(n_clips):
        fac.make(fpc, seed + 1000 * c, out=allf[c * fpc:(c + 1) * fpc])
    torch.cuda.synchronize()
    return allf


def test_c4_one_gpus_share_at_full_size(gpu_vs, oracle):
    """configs[3] as one of 8 GPUs sees it: 8 clips x 120 frames of 1080p, resident in HBM, through ONE
    vs_aligner_align_clips call.  Every clip starts with a first frame, every other frame aligns, a clip's results do not depend on its
    neighbours -- clips 0, 3 and 7 re-run alone through a fresh aligner give the same transforms bit for bit -- and ALL 960 frames equal
    one oracle aligner per clip (status, failure reason, iterations per level, transform within 1e-4)."""
    n_clips, fpc, w, h = 8, 120, 1920, 1080
    allf = _device_clips(w, h, n_clips, fpc, seed=1000)
    kw = dict(pyramid_min_width=256)
    al = gpu_vs.Aligner(device=0, **kw)
    st, ts = al.align_clips(n_clips * fpc, n_clips, mem_ptr=allf.data_ptr(), w=w, h=h, fmt=gpu_vs.FMT_BGR8)
    assert [st[c * fpc] for c in range(n_clips)] == [0] * n_clips
    assert sum(st) == n_clips * (fpc - 1)
    for c in (0, 3, 7):
        alone = gpu_vs.Aligner(device=0, **kw)
        s1, t1 = alone.align_batch_device(allf[c * fpc].data_ptr(), fpc, w, h, gpu_vs.FMT_BGR8)
        assert list(s1) == list(st[c * fpc:(c + 1) * fpc])
        assert [t.tup() for t in t1] == [t.tup() for t in ts[c * fpc:(c + 1) * fpc]], c
    for c in range(n_clips):
        cpu = oracle.Aligner(**kw)                          # a fresh VideoAligner per clip, as grid_search_align.cpp:174 does
        host = allf[c * fpc:(c + 1) * fpc].cpu().numpy()
        for k in range(fpc):
            ok_c, t_c = cpu.align_next(host[k])
            _same_alignment(al.info(c * fpc + k), cpu.debug(), st[c * fpc + k], ok_c, ts[c * fpc + k], t_c, (c, k))


def test_c5_one_gpus_share_at_full_size(gpu_vs):
    """configs[4] as one of 8 GPUs sees it: 8 clips x 60 frames of 4K 10-bit through the full stabilizer loop in ONE
    vs_stabilizer_process_clips call on device memory (24 GB in, 23 GB out).  Properties: `lag` frames of latency per clip,
    the 10-bit range is kept, and clip 5 re-run alone through a fresh stabilizer gives the same pixels bit for bit."""
    import torch
    n_clips, fpc, w, h, lag, crop = 8, 60, 3840, 2160, 10, 32
    allf = _device_clips(w, h, n_clips, fpc, seed=2000, bits=10)
    out = torch.zeros((n_clips * fpc, h - 2 * crop, w - 2 * crop, 3), dtype=torch.int16, device=allf.device)
    s = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_LANCZOS2, pyramid_min_width=256)
    r, has = s.process_clips_device(allf.data_ptr(), n_clips, fpc, w, h, gpu_vs.FMT_BGR10, out.data_ptr())
    assert has == ([0] * lag + [1] * (fpc - lag)) * n_clips and r == n_clips * (fpc - lag)
    produced = out.view(n_clips, fpc, h - 2 * crop, w - 2 * crop, 3)[:, lag:]
    assert int(produced.max()) <= 1023 and int(produced.min()) >= 0
    c = 5
    alone = gpu_vs.Stabilizer(device=0, warp_mode=gpu_vs.WARP_LANCZOS2, pyramid_min_width=256)
    out1 = torch.zeros((fpc, h - 2 * crop, w - 2 * crop, 3), dtype=torch.int16, device=allf.device)
    r1, has1 = alone.process_batch_device(allf[c * fpc].data_ptr(), fpc, w, h, gpu_vs.FMT_BGR10, out1.data_ptr())
    assert has1 == has[:fpc] and r1 == fpc - lag
    assert torch.equal(out1[lag:], out[c * fpc + lag:(c + 1) * fpc])


@pytest.mark.parametrize("bits,mode", [(8, 0), (10, 2)])
def test_overlapped_clip_groups_equal_clip_by_clip(gpu_vs, bits, mode):
    """vs_stabilizer_process_clips on dense device-resident clips cuts the batch into clip groups and runs the warps of group g
    on a stream of their own under the alignment of group g + 1 (small-footprint solver build).  7 clips x 12 frames -> groups of
    3 / 3 / 1 clips: outputs and flags must equal every clip run through a fresh stabilizer on its own, bit for bit."""
    torch = pytest.importorskip("torch")
    from video_stabilizer_amd import synth
    dev = torch.device("cuda", 0)
    w, h, fpc, n_clips = 640, 360, 12, 7
    factory = synth.TorchClipFactory(w, h, 4100, dev, channels=3, bits=bits)
    dt = torch.uint8 if bits == 8 else torch.int16
    allf = torch.empty((n_clips * fpc, h, w, 3), dtype=dt, device=dev)
    for c in range(n_clips):
        factory.make(fpc, 4100 + c, out=allf[c * fpc:(c + 1) * fpc])
    torch.cuda.synchronize()
    fmt = gpu_vs.FMT_BGR8 if bits == 8 else gpu_vs.FMT_BGR10
    kw = dict(lag=3, smoother_memory=2, crop_pixels=8, warp_mode=mode)
    out = torch.zeros((n_clips * fpc, h - 16, w - 16, 3), dtype=dt, device=dev)
    r, has = gpu_vs.Stabilizer(device=0, **kw).process_clips_device(allf.data_ptr(), n_clips, fpc, w, h, fmt, out.data_ptr())
    assert r == n_clips * (fpc - 3) and sum(has) == r
    for c in range(n_clips):
        one = torch.zeros((fpc, h - 16, w - 16, 3), dtype=dt, device=dev)
        r1, has1 = gpu_vs.Stabilizer(device=0, **kw).process_batch_device(allf[c * fpc].data_ptr(), fpc, w, h, fmt, one.data_ptr())
        assert has1 == has[c * fpc:(c + 1) * fpc], c
        assert torch.equal(one, out[c * fpc:(c + 1) * fpc]), c
    # a second call on the same handle (streams, events and scratch reused) gives the same bytes
    st = gpu_vs.Stabilizer(device=0, **kw)
    out2 = torch.zeros_like(out)
    for _ in range(2):
        out2.zero_()
        st.process_clips_device(allf.data_ptr(), n_clips, fpc, w, h, fmt, out2.data_ptr())
    assert torch.equal(out2, out)


@pytest.mark.parametrize("n,bits,n2", [(120, 8, 30), (101, 10, 30), (100, 8, 97)])
def test_long_device_clip_in_time_chunks_equals_frame_by_frame(gpu_vs, n, bits, n2):
    """vs_stabilizer_process_batch on ONE long dense device-resident clip is cut in time (chunks of >= 48 frames): the warps of
    chunk c run on their own stream under the alignment of chunk c + 1, frames queued across a chunk boundary stay pointers into
    the caller's batch.  Must equal n successive process() calls bit for bit -- and a second batch on the same handle (frames of
    the first batch still queued in buffers of the handle's own) must continue the sequence exactly as the calls do, whether that
    second batch is one piece (30 frames) or cut in time itself (97 frames: its first warps read the handle's own buffers on the
    second stream)."""
    torch = pytest.importorskip("torch")
    from video_stabilizer_amd import synth
    dev = torch.device("cuda", 0)
    w, h = 480, 270
    dt = torch.uint8 if bits == 8 else torch.int16
    clip, _ = synth.TorchClipFactory(w, h, 4200, dev, channels=3, bits=bits).make(n + n2, 4201)
    torch.cuda.synchronize()
    fmt = gpu_vs.FMT_BGR8 if bits == 8 else gpu_vs.FMT_BGR10
    kw = dict(lag=5, smoother_memory=3, crop_pixels=8, warp_mode=gpu_vs.WARP_LANCZOS2)
    host = clip.cpu().numpy()
    if bits != 8:
        host = host.view(np.uint16)
    seq = gpu_vs.Stabilizer(device=0, **kw)
    want = [seq.process(f, fmt=fmt) for f in host]
    bat = gpu_vs.Stabilizer(device=0, **kw)
    out = torch.zeros((n + n2, h - 16, w - 16, 3), dtype=dt, device=dev)
    r1, has1 = bat.process_batch_device(clip[0].data_ptr(), n, w, h, fmt, out.data_ptr())                    # time chunks
    r2, has2 = bat.process_batch_device(clip[n].data_ptr(), n2, w, h, fmt, out[n].data_ptr())                 # continues the clip
    assert r1 == n - 5 and r2 == n2
    got = out.cpu().numpy()
    if bits != 8:
        got = got.view(np.uint16)
    for i, wnt in enumerate(want):
        assert bool((has1 + has2)[i]) == (wnt is not None), i
        if wnt is not None:
            assert np.array_equal(got[i], wnt), i
