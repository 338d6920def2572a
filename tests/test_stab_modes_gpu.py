"""The stabilizer's scheduling machinery cannot change a byte.  Device-resident vs_stabilizer_process_clips / _batch run by default with
the warps on a second stream, the next chunk's alignment prefetched, deferred buffer release and the small-footprint solver build
(vs_engine.hip stab_run); every piece has a switch that is read ONCE per process (VS_STAB_OVERLAP, VS_STAB_PREFETCH,
VS_GN_CORESIDENT), so each combination runs in a child process and prints a digest of its outputs: all must be equal -- to each other
and to the frame-by-frame calls.  VS_GN_SELECT_DEPTH=2 on top sends every pair through the "libstdc++ would have heap-selected" exit
(fail_reason 100 -> the chunk is redone through the per-level host path) while a prefetched chunk is in flight: still the same bytes.
Round 5: with the library's DEFAULT warp (VS_WARP_BILINEAR_CV) the overlapped chunks keep the exclusive solver build, clip batches are cut into two
groups and time chunks hold at least 120 frames (VS_STAB_CV_SOLVER=1: the small build there too) -- the same comparison for that configuration.
"""
import hashlib
import itertools
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
from video_stabilizer_amd import capi, synth
W, H, crop = 320, 240, 8
n_clips, fpc = 3, 40                                           # 39 pairs per clip: the small solver build's threshold is 32
frames = np.concatenate([synth.make_clip(W, H, fpc, seed=50 + c, channels=3)[0] for c in range(n_clips)])
import os
cv = os.environ.get("VS_T_WARP") == "cv"                       # the library defaults (fixed-point bilinear, constant border) instead of the contracted Lanczos2
kw = {} if cv else dict(warp_mode=capi.WARP_LANCZOS2_FAST)
reps = 2 if cv else 1                                          # (that warp's time chunks hold >= 120 frames: the long sequence is the clips twice over)
dev = torch.from_numpy(np.concatenate([frames] * reps)).cuda()
out = torch.zeros((reps * n_clips * fpc, H - 2 * crop, W - 2 * crop, 3), dtype=torch.uint8, device="cuda")
st = capi.Stabilizer(device=0, lag=6, crop_pixels=crop, **kw)
r, has = st.process_clips_device(dev.data_ptr(), n_clips, fpc, W, H, capi.FMT_BGR8, out.data_ptr())
torch.cuda.synchronize()
h1 = hashlib.sha256(out[:n_clips * fpc].cpu().numpy().tobytes() + bytes(has)).hexdigest()
# one long clip in time chunks (>= 96 frames): the three clips back to back as ONE sequence
out.zero_()
st2 = capi.Stabilizer(device=0, lag=6, crop_pixels=crop, **kw)
r2, has2 = st2.process_batch_device(dev.data_ptr(), reps * n_clips * fpc, W, H, capi.FMT_BGR8, out.data_ptr())
torch.cuda.synchronize()
h2 = hashlib.sha256(out.cpu().numpy().tobytes() + bytes(has2)).hexdigest()
print("DIGEST", r, h1, r2, h2)
'''


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    return int(line[1]), line[2], int(line[3]), line[4]


def test_every_scheduling_switch_combination_gives_the_same_bytes(gpu_vs):
    from video_stabilizer_amd import synth
    W, H, crop, n_clips, fpc = 320, 240, 8, 3, 40
    frames = np.concatenate([synth.make_clip(W, H, fpc, seed=50 + c, channels=3)[0] for c in range(n_clips)])
    # the reference point: every clip through its own stabilizer, one process call per frame
    outs, has = [], []
    for c in range(n_clips):
        st = gpu_vs.Stabilizer(device=0, lag=6, crop_pixels=crop, warp_mode=gpu_vs.WARP_LANCZOS2_FAST)
        for f in frames[c * fpc:(c + 1) * fpc]:
            o = st.process(f)
            has.append(1 if o is not None else 0)
            outs.append(o if o is not None else np.zeros((H - 2 * crop, W - 2 * crop, 3), np.uint8))
    want_clips = hashlib.sha256(np.stack(outs).tobytes() + bytes(has)).hexdigest()
    st = gpu_vs.Stabilizer(device=0, lag=6, crop_pixels=crop, warp_mode=gpu_vs.WARP_LANCZOS2_FAST)
    outs, has = [], []
    for f in frames:
        o = st.process(f)
        has.append(1 if o is not None else 0)
        outs.append(o if o is not None else np.zeros((H - 2 * crop, W - 2 * crop, 3), np.uint8))
    want_seq = hashlib.sha256(np.stack(outs).tobytes() + bytes(has)).hexdigest()

    seen = {}
    for ov, pf, co in itertools.product("01", "01", "01"):
        seen[ov, pf, co, "-"] = _run({"VS_STAB_OVERLAP": ov, "VS_STAB_PREFETCH": pf, "VS_GN_CORESIDENT": co})
    # the depth-limit exit (host redo of the chunk) with and without a prefetched chunk in flight
    seen["1", "1", "-", "2"] = _run({"VS_STAB_OVERLAP": "1", "VS_STAB_PREFETCH": "1", "VS_GN_SELECT_DEPTH": "2"})
    seen["1", "0", "-", "2"] = _run({"VS_STAB_OVERLAP": "1", "VS_STAB_PREFETCH": "0", "VS_GN_SELECT_DEPTH": "2"})
    for key, (r, h1, r2, h2) in seen.items():
        assert r == n_clips * (fpc - 6) and h1 == want_clips, ("clips", key)
        assert r2 == n_clips * fpc - 6 and h2 == want_seq, ("one long clip", key)


def test_the_default_warps_scheduling_gives_the_same_bytes_too(gpu_vs):
    from video_stabilizer_amd import synth
    W, H, crop, n_clips, fpc = 320, 240, 8, 3, 40
    frames = np.concatenate([synth.make_clip(W, H, fpc, seed=50 + c, channels=3)[0] for c in range(n_clips)])

    def frame_by_frame(seq, restart_every):
        outs, has, st = [], [], None
        for i, f in enumerate(seq):
            if i % restart_every == 0:
                st = gpu_vs.Stabilizer(device=0, lag=6, crop_pixels=crop)                  # library defaults: VS_WARP_BILINEAR_CV, constant border
            o = st.process(f)
            has.append(1 if o is not None else 0)
            outs.append(o if o is not None else np.zeros((H - 2 * crop, W - 2 * crop, 3), np.uint8))
        return hashlib.sha256(np.stack(outs).tobytes() + bytes(has)).hexdigest()
    want_clips = frame_by_frame(frames, fpc)
    want_seq = frame_by_frame(np.concatenate([frames, frames]), 1 << 30)
    for env in ({}, {"VS_STAB_CV_SOLVER": "1"}, {"VS_STAB_OVERLAP": "0"}):
        r, h1, r2, h2 = _run(dict(env, VS_T_WARP="cv"))
        assert r == n_clips * (fpc - 6) and h1 == want_clips, ("clips", env)
        assert r2 == 2 * n_clips * fpc - 6 and h2 == want_seq, ("one long clip", env)
