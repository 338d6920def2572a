#!/usr/bin/env python3
"""Phase table of the tuned bgr_image_warp kernel from in-kernel time stamps (analysis build: tools/build_variant.sh stamps
"-DVS_WARP_STAMPS=1"; run with VS_AMD_LIB=video_stabilizer_amd/variants/libvs_amd_stamps.so).  Every wave of the first 8192 interior
workgroups of the last launch leaves s_memtime at: entry, geometry done, loads issued, loads landed (an added s_waitcnt vmcnt(0)),
tile written, barrier passed, rows stored, stores drained -- and its HW_ID / XCC_ID, which say what shared a CU with it.
usage: python tools/warp_stamps.py [--mode bilinear --w 3840 --h 2160 --frames 4]"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--h", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--mode", default="bilinear")
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--transform", default="0.002,-0.0015,3.3,-2.7")
    args = ap.parse_args()
    import torch
    from video_stabilizer_amd import capi
    L = capi.lib()
    if not hasattr(L, "vs_debug_warp_stamps"):
        sys.exit("Error: this library has no stamps (build the 'stamps' variant and point VS_AMD_LIB at it)")
    L.vs_debug_warp_stamps.restype = ctypes.c_int
    L.vs_debug_warp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    dev = torch.device("cuda", 0)
    n, W, H = args.frames, args.w, args.h
    dt = torch.uint8 if args.bits == 8 else torch.int16
    src = torch.randint(0, 256 if args.bits == 8 else 1024, (n, H, W, 3), device=dev, dtype=torch.int32).to(dt)
    dst = torch.empty_like(src)
    tr = [float(v) for v in args.transform.split(",")]
    ts = [capi.Transform.of(tr[0], tr[1], tr[2] + 0.37 * i, tr[3] - 0.21 * i) for i in range(n)]
    mode = {"lanczos2": capi.WARP_LANCZOS2, "bilinear": capi.WARP_BILINEAR, "fast": capi.WARP_LANCZOS2_FAST, "sep": capi.WARP_LANCZOS2_SEP, "cv": capi.WARP_BILINEAR_CV}[args.mode]
    st = torch.cuda.current_stream()

    def run():
        capi.bgr_image_warp_batch_device(src.data_ptr(), n, W, H, 3, args.bits, ts, dst.data_ptr(), mode, capi.BORDER_CLAMP,
                                         max_value=255 if args.bits == 8 else 1023, stream=st.cuda_stream)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:                       # settle the shader clock
        for _ in range(8):
            run()
        torch.cuda.synchronize()
    NW, NS = 8192, 10
    buf = np.zeros(NW * 4 * NS, dtype=np.uint64)
    assert L.vs_debug_warp_stamps(buf.ctypes.data, buf.size) == NW          # discard what the settling runs left
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    run()
    b.record(st)
    torch.cuda.synchronize()
    launch_us = a.elapsed_time(b) * 1e3
    assert L.vs_debug_warp_stamps(buf.ctypes.data, buf.size) == NW
    s = buf.reshape(NW, 4, NS).astype(np.int64)
    ok = s[:, :, 0].min(axis=1) > 0                              # workgroups that recorded (interior tiles)
    s = s[ok]
    t = s[:, :, :8]
    names = ["geometry", "issue loads", "wait for loads", "convert + write tile", "wait at barrier", "sample + store rows", "drain stores"]
    d = np.diff(t, axis=2)                                        # (wg, wave, 7)
    life = t[:, :, 7] - t[:, :, 0]
    # clock: cycles of the whole launch's recorded span against the event time is not available per XCD; report cycles
    out = {"mode": args.mode, "w": W, "h": H, "frames": n, "launch_us": round(launch_us, 1), "us_per_frame": round(launch_us / n, 2),
           "workgroups_recorded": int(ok.sum()), "phases_cycles_mean_per_wave": {}, "phases_cycles_median_per_wave": {}}
    for i, nm in enumerate(names):
        out["phases_cycles_mean_per_wave"][nm] = round(float(d[:, :, i].mean()), 1)
        out["phases_cycles_median_per_wave"][nm] = round(float(np.median(d[:, :, i])), 1)
    out["wave_lifetime_cycles_mean"] = round(float(life.mean()), 1)
    out["wave_lifetime_cycles_median"] = round(float(np.median(life)), 1)
    # what shares a CU: key = XCC_ID[3:0], HW_ID[15:8] (cu, sh, se)
    hw = s[:, 0, 8]
    xcc = s[:, 0, 9] & 0xf
    key = (xcc << 8) | ((hw >> 8) & 0xff)
    wg0 = t[:, :, 0].min(axis=1)
    wg1 = t[:, :, 7].max(axis=1)
    conc, gaps, per_cu, mid = [], [], [], []
    for k in np.unique(key):
        m = key == k
        b0, e0 = wg0[m], wg1[m]
        span = e0.max() - b0.min()
        if span <= 0 or m.sum() < 4:
            continue
        conc.append((e0 - b0).sum() / span)
        lo, hi = b0.min() + 0.25 * span, b0.min() + 0.75 * span     # steady state: the middle half of this CU's recorded span
        mid.append((np.clip(e0, lo, hi) - np.clip(b0, lo, hi)).sum() / (hi - lo))
        per_cu.append(int(m.sum()))
        # busy-slot view: a new workgroup can start when one ends; the delay from the k-th end to the (k + resident)-th start
        o = np.sort(b0)
        gaps.append(np.median(np.diff(o)))
    out["cus_seen"] = len(conc)
    out["workgroups_per_cu_recorded_mean"] = round(float(np.mean(per_cu)), 1)
    out["resident_workgroups_per_cu_mean"] = round(float(np.mean(conc)), 2)
    out["resident_workgroups_per_cu_steady_state"] = round(float(np.mean(mid)), 2)
    out["median_cycles_between_workgroup_starts_on_a_cu"] = round(float(np.median(gaps)), 1)
    out["workgroup_lifetime_cycles_mean"] = round(float((wg1 - wg0).mean()), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
