"""Distinct handles are independent (the reference runs one instance per thread, grid_search_align.cpp:174): six threads, each with its own
aligner AND stabilizer, every entry-point family at once -- frame at a time, batches from host and device memory, clips, warps through the shared
parameter ring -- against the same calls made one after the other.  Bit for bit, three rounds."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_SCALE = max(1, int(os.environ.get("VS_SWEEP_SCALE", "1")))


def _job(G, torch, k, clip, dev):
    """everything thread k does; returns a flat list of comparable results"""
    out = []
    kw = dict(pyramid_min_width=32 + 4 * k, pyramid_min_height=24)
    al = G.Aligner(device=0, select_mode=k % 3, **kw)
    out.append([(ok, t.tup()) for ok, t in (al.align_next(f) for f in clip)])
    st, ts = G.Aligner(device=0, select_mode=k % 3, **kw).align_batch(clip)
    out.append((st, [t.tup() for t in ts]))
    n, h, w = clip.shape[:3]
    st, ts = G.Aligner(device=0, select_mode=k % 3, **kw).align_batch_device(dev.data_ptr(), n, w, h, G.FMT_BGR8)
    out.append((st, [t.tup() for t in ts]))
    s = G.Stabilizer(device=0, lag=2, crop_pixels=4 + k, warp_mode=k % 5, **kw)                     # all five warp modes over the six threads
    o, has = s.process_batch(clip)
    out.append((has, o.tobytes()))
    s2 = G.Stabilizer(device=0, lag=1 + k % 3, crop_pixels=2, warp_mode=(k + 1) % 5, **kw)
    out.append([None if f is None else f.tobytes() for f in (s2.process(f) for f in clip)])
    tr = [G.Transform.of(0.001 * k, -0.002, 1.5 + k, -0.5 * i) for i in range(n)]
    out.append(G.bgr_image_warp_batch(clip, tr, (k + 2) % 5, k % 2).tobytes())
    return out


def test_six_threads_with_their_own_handles_equal_the_serial_run(gpu_vs):
    import torch
    from video_stabilizer_amd import synth
    K = 6
    clips = [synth.make_clip(256 + 16 * k, 160 + 8 * k, 9, seed=700 + k, channels=3)[0] for k in range(K)]
    devs = [torch.from_numpy(c).to("cuda:0") for c in clips]
    serial = [_job(gpu_vs, torch, k, clips[k], devs[k]) for k in range(K)]
    for rnd in range(3 * _SCALE):
        got, err = [None] * K, [None] * K

        def work(k):
            try:
                got[k] = _job(gpu_vs, torch, k, clips[k], devs[k])
            except Exception as e:                              # noqa: BLE001 (reported below, with its thread)
                err[k] = e
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert err == [None] * K, err
        for k in range(K):
            assert got[k] == serial[k], (rnd, k)
