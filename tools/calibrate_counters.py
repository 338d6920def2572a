#!/usr/bin/env python3
"""Known-byte-count run for rocprofv3 FETCH_SIZE / WRITE_SIZE in the warp kernel's access width (12 B/lane).
Run under `rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...`; expected bytes are printed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi

n = 1536 * 1024 * 1024   # 1.5 GiB, well past the 256 MiB Infinity Cache
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    capi._check(capi.lib().vs_calib_copy12(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, None))
torch.cuda.synchronize()
assert torch.equal(src, dst)
print("calib_copy12 bytes per launch: read %d write %d (KiB %d)" % (n, n, n // 1024))
