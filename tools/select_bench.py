#!/usr/bin/env python3
"""Latency of the on-device nth_element replica (one workgroup per array; the launch time of <= 256 arrays is the
per-array latency).  Device-resident inputs."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_stabilizer_amd import capi

out = {}
for (tx, ty) in ((96, 54), (48, 27), (192, 108)):
    n = 200
    wd = torch.poisson(torch.full((n, ty, tx), 3.0, device="cuda")).to(torch.int16)
    idx = torch.empty((n, ty * tx), dtype=torch.int32, device="cuda")
    st = torch.empty((n,), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream()
    def run():
        capi._check(capi.lib().vs_select_smallest(C.c_void_p(wd.data_ptr()), n, tx, ty, float(os.environ.get("VS_FRAC", "0.8")), C.c_void_p(idx.data_ptr()),
                                                  C.c_void_p(st.data_ptr()), capi.MEM_DEVICE, C.c_void_p(s.cuda_stream)))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(10):
        run()
    b.record(s)
    torch.cuda.synchronize()
    out["%dx%d" % (tx, ty)] = round(a.elapsed_time(b) / 10 * 1e3, 1)
print(json.dumps({"select_us_per_array": out}))
