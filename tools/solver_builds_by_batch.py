#!/usr/bin/env python3
"""alignment alone (no warp): the two solver builds by batch size -- clips of 120 frames, 1080p, device resident"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth
dev = torch.device("cuda", 0)
W, H, n = 1920, 1080, 120
fac = synth.TorchClipFactory(W, H, 5, dev, channels=3, bits=8)
for n_clips in (2, 4, 8, 16):
    fr = torch.empty((n_clips * n, H, W, 3), dtype=torch.uint8, device=dev)
    for c in range(n_clips):
        fac.make(n, 100 + c, out=fr[c * n:(c + 1) * n])
    torch.cuda.synchronize()
    res = {}
    for mode, name in ((capi.BATCH_EXCLUSIVE, "exclusive"), (capi.BATCH_SHARED, "shared")):
        al = capi.Aligner(device=0, pyramid_min_width=256)
        al.set_batch_mode(mode)
        al.align_clips(n_clips * n, n_clips, mem_ptr=fr.data_ptr(), w=W, h=H, fmt=capi.FMT_BGR8, raw=True)
        al.enable_timing(True)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            st, ts = al.align_clips(n_clips * n, n_clips, mem_ptr=fr.data_ptr(), w=W, h=H, fmt=capi.FMT_BGR8, raw=True)
        dt = (time.perf_counter() - t0) / reps
        tm = al.timings()
        res[name] = (round(1e3 * dt, 3), round(tm["gn"]["ms"] / reps, 3), [t.tup() for t in ts][:3])
    same = res["exclusive"][2] == res["shared"][2]
    print("clips", n_clips, "pairs", n_clips * (n - 1), "| exclusive: ms/call %.3f gn %.3f | shared: ms/call %.3f gn %.3f | same %s" %
          (res["exclusive"][0], res["exclusive"][1], res["shared"][0], res["shared"][1], same), flush=True)
    del fr
