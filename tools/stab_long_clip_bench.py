#!/usr/bin/env python3
"""one long device-resident clip through vs_stabilizer_process_batch: time chunks overlapped (default) vs VS_STAB_OVERLAP=0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth
dev = torch.device("cuda", 0)
for (W, H, n) in ((1920, 1080, 480), (3840, 2160, 240)):
    clip, _ = synth.TorchClipFactory(W, H, 7, dev, channels=3, bits=8).make(n, 8)
    out = torch.empty((n, H - 64, W - 64, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for name, mode in (("bilinear_cv (the default)", capi.WARP_BILINEAR_CV), ("separable", capi.WARP_LANCZOS2_SEP), ("exact", capi.WARP_LANCZOS2)):
        st = capi.Stabilizer(device=0, warp_mode=mode, pyramid_min_width=256)
        best = 1e9
        for rep in range(5):
            st.reset()
            t0 = time.perf_counter()
            r, has = st.process_batch_device(clip.data_ptr(), n, W, H, capi.FMT_BGR8, out.data_ptr())
            best = min(best, time.perf_counter() - t0)
        print("VS_STAB_OVERLAP=%s VS_STAB_PREFETCH=%s VS_STAB_CV_SOLVER=%s %dx%d x%d, %s warp: %.2f ms per batch, %.0f frames/s, outputs %d"
              % (os.environ.get("VS_STAB_OVERLAP", "1"), os.environ.get("VS_STAB_PREFETCH", "1"), os.environ.get("VS_STAB_CV_SOLVER", "default"), W, H, n, name, 1e3 * best, n / best, r), flush=True)
    del clip, out
