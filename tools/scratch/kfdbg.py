import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
from video_stabilizer_amd import capi
from oracle import oracle
rng = np.random.default_rng(5)
for (w, h) in [(640, 480)]:
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ts, lmx, lmy, jx, jy = capi.keyframe_fused(img)
    gx, gy = oracle.grad_xy(img)
    ots, olx, oly = oracle.grad_argmax(gx, gy)
    good = ~((lmx != olx).any(axis=0))
    print("good rows:", np.nonzero(good.any(axis=1))[0])
    for r in np.nonzero(good.any(axis=1))[0][:6]:
        print(r, "".join("#" if g else "." for g in good[r]))
    print("nonzero got rows:", np.nonzero((lmx != 0).any(axis=(0, 2)))[0])
    r = np.nonzero((lmx != 0).any(axis=(0, 2)))[0]
    for rr in r[:6]:
        print(rr, lmx[0, rr, :12], lmx[1, rr, :12], "want", olx[0, rr, :12], olx[1, rr, :12])
