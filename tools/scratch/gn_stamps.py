import os, sys
sys.path.insert(0, "/root/repo")
import torch
from video_stabilizer_amd import capi, synth
W, H, n = 1920, 1080, 6
frames, _ = synth.make_clip_torch(W, H, n, seed=5, device=torch.device("cuda", 0))
torch.cuda.synchronize()
al = capi.Aligner(device=0, pyramid_min_width=256)
for i in range(n):
    sys.stderr.write(f"--- frame {i}\n"); sys.stderr.flush()
    al.align_batch_device(frames[i].data_ptr(), 1, W, H, capi.FMT_BGR8)
