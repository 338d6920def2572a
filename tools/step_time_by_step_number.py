#!/usr/bin/env python3
"""Per-step warp-launch time of bench.py's c2 step as a function of the step number (is the timed region of `--steps 20 --warmup 5`
at steady-state clocks?), after the clip generation and after an idle second."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from video_stabilizer_amd import capi, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = types.SimpleNamespace(select="device", no_warp=False, exclusive_solver=False, warp_mode="exact")
wl = bench.WORKLOADS["c2"]
aw = bench.AlignWarp(torch, capi, synth, dev, wl, wl["frames"], 1, [wl["seed"]], dict(pyramid_min_width=256), args, wl["seed"])
for label, idle in (("right after clip generation", 0.0), ("after 1 s idle", 1.0), ("after 5 s idle", 5.0)):
    torch.cuda.synchronize()
    time.sleep(idle)
    aw.ev = []
    t0 = time.perf_counter()
    for i in range(80):
        aw.step(True)
    torch.cuda.synchronize()
    total = 1e3 * (time.perf_counter() - t0)
    d = [a.elapsed_time(b) for a, b in aw.ev]
    print(label, "80 steps %.1f ms = %.3f per step; warp launch ms per step:" % (total, total / 80), " ".join("%.2f" % x for x in d[:12]), "| 12-24: %.3f  25-44: %.3f  45-79: %.3f" % (sum(d[12:25]) / 13, sum(d[25:45]) / 20, sum(d[45:]) / 35), flush=True)
