import sys, time, json
sys.path.insert(0, '/root/repo')
import torch
from video_stabilizer_amd import capi, synth
W, H, n = 1920, 1080, 64
frames, _ = synth.make_clip_torch(W, H, n, seed=5, device=torch.device("cuda", 0))
torch.cuda.synchronize()
al = capi.Aligner(device=0, pyramid_min_width=256)
for rep in range(2):
    al.reset(); al.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(n):
        al.align_batch_device(frames[i].data_ptr(), 1, W, H, capi.FMT_BGR8)
    dt = time.perf_counter() - t0
tm = al.timings()
print(json.dumps({"ms_per_frame": 1e3*dt/n, "stages_ms_per_frame": {k: round(v["ms"]/n, 4) for k, v in tm.items() if isinstance(v, dict)}, "iters": tm["gn_iterations"]/n}))
