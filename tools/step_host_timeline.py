#!/usr/bin/env python3
"""Host-side timeline of bench.py's c2 step loop: how long the host spends inside align() and inside warp() per step, and per-step GPU warp time
(events).  Question: what makes the alignment chains run three-under-one-warp and then not at all for three warps (profiles/r05_step_trace_shared.md)?"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from video_stabilizer_amd import capi, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
mode_name = sys.argv[1] if len(sys.argv) > 1 else "separable"
args = types.SimpleNamespace(select="device", no_warp=False, exclusive_solver=False, warp_mode=mode_name)
wl = bench.WORKLOADS["c2"]
aw = bench.AlignWarp(torch, capi, synth, dev, wl, wl["frames"], 1, [wl["seed"]], dict(pyramid_min_width=256), args, wl["seed"])
mode = getattr(capi, bench.WARP_MODES[mode_name][0])
for _ in range(60):
    aw.step(False)
torch.cuda.synchronize()
rows = []
aw.ev = []
t00 = time.perf_counter()
for i in range(24):
    t0 = time.perf_counter()
    status, ts = aw.align()
    t1 = time.perf_counter()
    aw.warp(ts, mode, True)
    t2 = time.perf_counter()
    rows.append((1e3 * (t0 - t00), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
torch.cuda.synchronize()
total = 1e3 * (time.perf_counter() - t00)
print("24 steps %.2f ms = %.3f per step" % (total, total / 24))
for i, ((ts_, a, w), (ea, eb)) in enumerate(zip(rows, aw.ev)):
    print("step %2d  host t=%7.2f ms  align() %.3f ms  warp() call %.3f ms  | GPU warp %.2f ms" % (i, ts_, a, w, ea.elapsed_time(eb)))
