#!/usr/bin/env python3
"""Small alignment-only driver for rocprofv3 --pmc passes (no torch, so the profiler sees one HIP runtime):
a synthetic 1080p clip, host frames, batched alignment a few times.
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d out -- python3 tools/align_pmc.py
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_stabilizer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--frames", type=int, default=48)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--unique", type=int, default=8, help="distinct synthetic frames (cycled; numpy generation is slow)")
ap.add_argument("--device-resident", action="store_true", help="frames copied to HBM first: the whole batch is one launch per stage")
args = ap.parse_args()
base, _ = synth.make_clip(args.w, args.h, args.unique, seed=3, channels=3)
frames = np.ascontiguousarray(np.concatenate([base[::(1 if (i // args.unique) % 2 == 0 else -1)] for i in range(0, args.frames, args.unique)])[:args.frames])
a = capi.Aligner(device=0, pyramid_min_width=256)
dptr = None
if args.device_resident:
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(frames.nbytes)) == 0
    assert hip.hipMemcpy(p, frames.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(frames.nbytes), 1) == 0
    dptr = p.value
for r in range(args.reps):
    a.reset()
    t0 = time.perf_counter()
    if dptr is not None:
        st, _ = a.align_batch_device(dptr, args.frames, args.w, args.h, capi.FMT_BGR8)
    else:
        st, _ = a.align_batch(frames)
    print("rep", r, "aligned", sum(st), "of", len(st), "%.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
