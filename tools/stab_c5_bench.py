#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU (8 x 60 x 4K 10-bit clips through vs_stabilizer_process_clips) with the library's default warp
(VS_WARP_BILINEAR_CV) and with the separable Lanczos2: frames/s, best of 4.  Knobs come from the environment (VS_STAB_GROUPS,
VS_STAB_OVERLAP, VS_STAB_CV_SOLVER, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth
dev = torch.device("cuda", 0)
W, H, n, nc, crop = 3840, 2160, 60, int(os.environ.get("C5_CLIPS", "8")), 32
f = synth.TorchClipFactory(W, H, 5, dev, channels=3, bits=10)
frames = torch.empty((nc * n, H, W, 3), dtype=torch.int16, device=dev)
for j in range(nc):
    f.make(n, 5 + 1000 * j, out=frames[j * n:(j + 1) * n])
out = torch.empty((nc * n, H - 2 * crop, W - 2 * crop, 3), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("VS_STAB"))
for name, kw in (("bilinear_cv (library default)", {}), ("separable lanczos2, clamp", dict(warp_mode=capi.WARP_LANCZOS2_SEP, warp_border=capi.BORDER_CLAMP))):
    st = capi.Stabilizer(device=0, pyramid_min_width=256, **kw)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        r = st.process_clips_device(frames.data_ptr(), nc, n, W, H, capi.FMT_BGR10, out.data_ptr())[0]
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("[%s] %d x %d x 4K 10-bit, %s: %.2f ms, %.0f frames/s, outputs %d" % (knobs, nc, n, name, 1e3 * best, nc * n / best, r), flush=True)
    del st
