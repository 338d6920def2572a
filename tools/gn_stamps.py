#!/usr/bin/env python3
"""Per-phase shader-clock stamps of one frame pair inside vs_k_align_pairs (diagnostic builds only):
  tools/build_variant.sh stamps "-DVS_PROFILE_STAMPS" vs_engine.hip     # level phases + the first two iterations
  tools/build_variant.sh pipe "-DVS_PIPE_STAMPS" vs_engine.hip          # per-wave times inside the pipelined iteration
  VS_AMD_LIB=video_stabilizer_amd/variants/libvs_amd_stamps.so python tools/gn_stamps.py
s_memtime counts at 2.4 GHz on MI355X (tools/ubench_clock.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi, synth
W, H, n = 1920, 1080, 6
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 0     # --batch 240: pair 0 of a full launch
frames, _ = synth.make_clip_torch(W, H, max(n, batch), seed=5, device=torch.device("cuda", 0))
torch.cuda.synchronize()
al = capi.Aligner(device=0, pyramid_min_width=256,
                  select_mode=capi.SELECT_STABLE if "--stable" in sys.argv else capi.SELECT_DEVICE)
if batch:
    for r in range(2):
        sys.stderr.write(f"--- batch of {batch}, pass {r}\n"); sys.stderr.flush()
        al.reset()
        al.align_batch_device(frames.data_ptr(), batch, W, H, capi.FMT_BGR8)
    sys.exit(0)
for i in range(n):
    sys.stderr.write(f"--- frame {i}\n"); sys.stderr.flush()
    al.align_batch_device(frames[i].data_ptr(), 1, W, H, capi.FMT_BGR8)
