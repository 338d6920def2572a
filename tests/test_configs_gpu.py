"""Every BASELINE.json config at its real frame size against the CPU oracle (configs[2..4]; configs[0..1] are in
test_engine_gpu.py).  Clips are made on the device by synth.TorchClipFactory (the generator bench.py uses) and copied
to the host once, so the GPU path and the oracle see identical bytes.

  C3  4K BGR: bgr_image_warp Lanczos2 of a whole frame == oracle, bit for bit; 4K BGR alignment (4 levels) == oracle
  C4  many independent 1080p clips through vs_aligner_align_clips == one oracle aligner per clip
  C5  10-bit BGR through the full stabilizer loop (vs_stabilizer_process_clips) == one oracle stabilizer per clip;
      one 3840x2160 10-bit clip of lag + 2 frames through size-independent properties
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
    assert g.params.warp_mode == (gpu_vs.WARP_BILINEAR if sampler == "default" else gpu_vs.WARP_LANCZOS2)
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
