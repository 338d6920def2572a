#!/usr/bin/env python3
"""Where one AlignNextFrame call (device-resident frame) spends its time: wall clock per call vs the per-stage device
times of vs_aligner_enable_timing.  usage: python tools/latency_stages.py [4k]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth

W, H, n = (1920, 1080, 48) if len(sys.argv) < 2 or sys.argv[1] != "4k" else (3840, 2160, 24)
frames, _ = synth.make_clip_torch(W, H, n, seed=5, device=torch.device("cuda", 0))
torch.cuda.synchronize()
al = capi.Aligner(device=0, pyramid_min_width=256)
out = {"w": W, "h": H}
for timing in (False, True):
    for _ in range(2):
        al.reset()
        al.enable_timing(timing)
        t0 = time.perf_counter()
        for i in range(n):
            al.align_batch_device(frames[i].data_ptr(), 1, W, H, capi.FMT_BGR8)
        dt = time.perf_counter() - t0
    key = "timed" if timing else "untimed"
    out[key] = {"ms_per_call": round(1e3 * dt / n, 4)}
    if timing:
        tm = al.timings()
        out[key]["stages_ms_per_call"] = {k: round(v["ms"] / n, 4) for k, v in tm.items() if isinstance(v, dict) and v["launches"]}
        out[key]["launches_per_call"] = {k: v["launches"] / n for k, v in tm.items() if isinstance(v, dict) and v["launches"]}
print(json.dumps(out))
