#!/usr/bin/env python3
"""Single-stream figures the batched bench does not show (DESIGN.md 'Measurement'):
  * sequential AlignNextFrame (one frame per call, device-resident frames): per-call latency, frames/s
  * the same with host-resident frames (PCIe-inclusive)
  * batched alignment fed from host memory (PCIe-inclusive batch rate)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth

W, H, n = (1920, 1080, 64) if len(sys.argv) < 2 or sys.argv[1] != "4k" else (3840, 2160, 32)
frames, _ = synth.make_clip_torch(W, H, n, seed=5, device=torch.device("cuda", 0))
host = frames.cpu().numpy()
torch.cuda.synchronize()
out = {"w": W, "h": H, "frames": n}
al = capi.Aligner(device=0, pyramid_min_width=256)
for _ in range(2):
    al.reset()
    t0 = time.perf_counter()
    ok = 0
    for i in range(n):
        st, _ = al.align_batch_device(frames[i].data_ptr(), 1, W, H, capi.FMT_BGR8)
        ok += st[0]
    dt = time.perf_counter() - t0
out["sequential_device_resident"] = {"ms_per_frame": round(1e3 * dt / n, 3), "fps": round(n / dt, 1), "aligned": ok}
for _ in range(2):
    al.reset()
    t0 = time.perf_counter()
    for i in range(n):
        al.align_next(host[i])
    dt = time.perf_counter() - t0
out["sequential_host_frames"] = {"ms_per_frame": round(1e3 * dt / n, 3), "fps": round(n / dt, 1)}
for _ in range(2):
    al.reset()
    t0 = time.perf_counter()
    al.align_batch(host)
    dt = time.perf_counter() - t0
out["batched_host_frames_pcie_inclusive"] = {"ms_per_frame": round(1e3 * dt / n, 3), "fps": round(n / dt, 1)}
print(json.dumps(out))
