#!/usr/bin/env python3
"""bench.py -- aligned frames/sec on synthetic video + bgr_image_warp HBM roofline (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3]

A "step" is one pass of the hot path over one clip that is already resident in HBM:
  c2 (default, BASELINE configs[1]): 240-frame 1080p BGR clip, pyramid_min_width=256 (3 levels);
      every frame is aligned to its predecessor (VideoAligner::AlignNextFrame semantics, batched)
      and then resampled with bgr_image_warp Lanczos2 by its measured transform (what the stabilizer
      does with each frame).
  c3 (BASELINE configs[2]): 120-frame 4K clip, 4 levels, same two stages.
value = frames aligned+warped per second, whole job (all ranks); one process per GPU, clips are
independent so ranks share nothing but the barrier and the max-over-ranks time ("weak" scaling).

The JSON line also carries
  roofline      the dominant kernel of the timed region (bgr_image_warp), algorithmic bytes / mean
                launch time measured with HIP events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline  the CPU restatement of the reference path (oracle/, kind "port") timed on the host cores
                on a bounded sample of the same clip
torch is used for device memory, streams, events and torch.distributed only.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    "c2": dict(name="1080p single clip, 3-level pyramid (pyramid_min_width=256), align + bgr_image_warp Lanczos2",
               w=1920, h=1080, frames=240, seed=1),
    "c3": dict(name="4K single clip, 4-level pyramid (pyramid_min_width=256), align + bgr_image_warp Lanczos2",
               w=3840, h=2160, frames=120, seed=2),
}


def cpu_baseline(frames_host, params_kw, seconds_budget=20.0):
    """oracle (CPU restatement of the reference) on the host cores: one aligner + warp per thread, each
    thread an independent copy of the sample (the reference's own multi-clip regime, grid_search_align.cpp:105-210)."""
    from oracle import oracle as O
    n = frames_host.shape[0]
    threads = max(1, min(os.cpu_count() or 1, 16))
    # calibrate on one thread to size the sample
    al = O.Aligner(**params_kw)
    t0 = time.perf_counter()
    al.align_next(frames_host[0])
    ok, t = al.align_next(frames_host[1])
    O.bgr_image_warp(frames_host[1], t)
    per_frame = (time.perf_counter() - t0) / 2
    sample = int(max(3, min(n, seconds_budget / max(per_frame, 1e-6))))
    done = [0] * threads

    def work(k):
        a = O.Aligner(**params_kw)
        for i in range(sample):
            ok, t = a.align_next(frames_host[i])
            O.bgr_image_warp(frames_host[i], t if ok else O.Transform.of())
            done[k] += 1

    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return {"value": round(sum(done) / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "first %d frames of the same clip, align + Lanczos2 warp, %d threads x 1 clip copy each, "
                      "CPU restatement of the Halide path (not Halide)" % (sample, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override the clip length")
    ap.add_argument("--select", default="device", choices=["host", "device"])
    ap.add_argument("--no-warp", action="store_true", help="alignment only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from video_stabilizer_amd import capi, synth
    from video_stabilizer_amd import dist as vsdist

    world, rank, local_rank = vsdist.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        # one process per GPU; "nccl" is RCCL on ROCm.  No data-path collective: clips are independent.
        dist = vsdist.init("nccl", rank, world, device_id=dev)

    wl = WORKLOADS[args.workload]
    W, H = wl["w"], wl["h"]
    n = args.frames or wl["frames"]
    frames, _ = synth.make_clip_torch(W, H, n, seed=wl["seed"] + 1000 * rank, device=dev, channels=3)
    warped = torch.empty_like(frames)
    torch.cuda.synchronize()

    params_kw = dict(pyramid_min_width=256)
    aligner = capi.Aligner(device=local_rank,
                           select_mode=capi.SELECT_DEVICE if args.select == "device" else capi.SELECT_STL_HOST, **params_kw)
    stream = torch.cuda.current_stream()
    ev = []   # (start, end) events around the warp launches of the timed steps

    def step(timed):
        aligner.reset()
        status, ts = aligner.align_batch_device(frames.data_ptr(), n, W, H, capi.FMT_BGR8)
        if not args.no_warp:
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
            capi.bgr_image_warp_batch_device(frames.data_ptr(), n, W, H, 3, 8, ts, warped.data_ptr(),
                                             capi.WARP_LANCZOS2, capi.BORDER_CLAMP, stream=stream.cuda_stream)
            if timed:
                b.record(stream)
                ev.append((a, b))
        return status

    def timed_loop(fn, k):
        """k calls of fn between barrier + synchronize on both sides; returns seconds on this rank"""
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            st = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0, st

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    aligner.enable_timing(True)
    dt, status = timed_loop(lambda: step(True), args.steps)
    # whole-job numbers: max seconds over ranks, frames summed over ranks (the only collectives of the run)
    dt, total_frames, total_aligned = vsdist.aggregate(dt, n * args.steps, int(sum(status)) * args.steps, device=dev)
    tm = aligner.timings()

    # second, separately reported figure: the alignment stage alone (BASELINE configs[1] read literally)
    def align_only():
        aligner.reset()
        return aligner.align_batch_device(frames.data_ptr(), n, W, H, capi.FMT_BGR8)[0]
    aligner.enable_timing(True)
    dt_a, _ = timed_loop(align_only, args.steps)
    dt_a, frames_a, _ = vsdist.aggregate(dt_a, n * args.steps, 0, device=dev)
    tm_a = aligner.timings()

    if rank == 0:
        def stage_table(t):
            return {k: {"ms_per_step": round(v["ms"] / args.steps, 4), "launches_per_step": v["launches"] // max(1, args.steps)}
                    for k, v in t.items() if isinstance(v, dict) and v["launches"]}
        stages = stage_table(tm)
        out = {
            "metric": "aligned frames/sec", "value": round(total_frames / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"], "frames_per_clip": n, "clips_per_gpu": 1, "clip_seeds": "rank r gets seed %d + 1000 r" % wl["seed"], "width": W, "height": H,
                       "selection": "std::nth_element on the host" if args.select == "host" else "on-device replica of libstdc++ nth_element (same survivors, same order)",
                       "warp": None if args.no_warp else "bgr_image_warp lanczos2 u8 clamp", "resident": "HBM"},
            "aligned_per_step": total_aligned // args.steps,
            "stages": stages,
            "gn_iterations_per_frame": round(tm["gn_iterations"] / max(1, tm["frames"]), 2),
            "align_only": {"value": round(frames_a / dt_a, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt_a / args.steps, 4),
                           "stages": stage_table(tm_a),
                           "note": "same clip, alignment stages only (no warp launch competing for the CUs)"},
        }
        if ev:
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)          # one launch per step, n frames per launch
            bytes_per_launch = W * H * 3 * 2 * n                          # W*H*3*(in+out) per frame
            achieved = bytes_per_launch / (ms * 1e-3) / 1e9
            # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json), scaled to
            # this launch's frame count; null when the profile for this frame size is not there
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                key = {"c2": "c2_1080p_240_frames", "c3": "c3_4k_32_frames"}[args.workload]
                per_frame = tj[key]["traffic_bytes"] / {"c2": 240, "c3": 32}[args.workload]
                traffic = int(per_frame * n)
            except Exception:
                pass
            out["roofline"] = {"kernel": "vs_k_bgr_warp_u8c3<lanczos2,clamp> (bgr_image_warp)", "bound": "hbm",
                               "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                               "launch_ms": round(ms, 4), "bytes_per_launch": bytes_per_launch,
                               "note": "VALU-issue-bound, not HBM-bound: ~270 VALU instructions per output pixel in the "
                                       "reference's exact fp32 order (DESIGN.md, profiles/r01_bgr_image_warp_pmc.md)"}
        if not args.no_cpu_baseline and world == 1:
            fh = frames[: min(n, 64)].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(fh, params_kw)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
