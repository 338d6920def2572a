#!/usr/bin/env python3
"""Does the kernel time depend on how long the GPU has been busy (clock ramp)?  bgr_image_warp 32 x 4K launched back to back for ~3 s,
per-launch HIP-event time printed as a function of elapsed busy time."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi
dev = torch.device("cuda", 0)
n, W, H = 32, 3840, 2160
src = torch.randint(0, 256, (n, H, W, 3), device=dev, dtype=torch.int32).to(torch.uint8)
dst = torch.empty_like(src)
ts = [capi.Transform.of(0.002, -0.0015, 3.3 + 0.37 * i, -2.7 - 0.21 * i) for i in range(n)]
st = torch.cuda.current_stream()
for mode, name in ((capi.WARP_LANCZOS2, "exact"), (capi.WARP_LANCZOS2_FAST, "contracted")):
    torch.cuda.synchronize()
    time.sleep(2.0)          # let the device fall idle
    evs = []
    t0 = time.perf_counter()
    for r in range(1500):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        capi.bgr_image_warp_batch_device(src.data_ptr(), n, W, H, 3, 8, ts, dst.data_ptr(), mode, capi.BORDER_CLAMP, max_value=255, stream=st.cuda_stream)
        b.record(st)
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    acc, out = 0.0, []
    marks = [0, 1, 2, 5, 10, 20, 40, 80, 160, 320, 640, 1000, 1499]
    for i, m in enumerate(ms):
        if i in marks: out.append((round(acc, 1), round(1e3 * m / n, 2)))
        acc += m
    print(name, "(busy ms so far, us per 4K frame):", out, flush=True)
