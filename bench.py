#!/usr/bin/env python3
"""bench.py -- aligned frames/sec on synthetic video + bgr_image_warp HBM roofline (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--clips-per-gpu C]

A "step" is one pass of the hot path over the rank's clips, which are already resident in HBM:
  c2 (default, BASELINE configs[1]): one 240-frame 1080p BGR clip, pyramid_min_width=256 (3 levels): every frame is
      aligned to its predecessor (VideoAligner::AlignNextFrame semantics, batched) and then resampled with
      bgr_image_warp Lanczos2 by its measured transform -- what the stabilizer does with each frame.
  c3 (configs[2]): one 120-frame 4K clip, 4 levels, same two stages.
  c4 (configs[3]): C clips x 120 frames 1080p per GPU (clip i -> rank i mod N), align + warp.
  c5 (configs[4]): C clips x 60 frames 4K 10-bit BGR per GPU through the full VideoStabilizer loop
      (lag 10, L1 smoother, decay, warp, crop) -- vs_stabilizer_process_batch.
value = frames per second, whole job (all ranks).  One process per GPU; clips are independent, so ranks share nothing
but the barriers that bracket the timed region and one max/sum all-reduce for the report ("weak" scaling).

The JSON line also carries
  roofline      the dominant kernel of the timed region (bgr_image_warp): algorithmic bytes / mean launch time, HIP
                events on the launch stream, against the 8 TB/s HBM peak; `traffic` scaled from the committed PMC passes
  roofline_4k   the same kernel where the north star quotes it: 32 x 4K frames per launch, isolated, after the timed
                region (exact and fast arithmetic), with the VALU / LDS busy fractions of the committed PMC passes --
                the kernel is VALU-issue-bound, the HBM fraction is what that leaves
  align_only    the same clip through the alignment stages alone (configs[1] read literally), with per-stage times
  cpu_baseline  the CPU restatement of the reference path (oracle/, kind "port") on the host cores, bounded sample
torch is used for device memory, streams, events and torch.distributed only.

--gpus N without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself: N child processes of this
script, one per GPU, spawned BEFORE this process touches the GPU (the pattern of the reference's only many-clip
precedent, the worker pool of grid_search_align.cpp:159-210); rank 0's JSON line is the output.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    "c2": dict(name="1080p single clip, 3-level pyramid (pyramid_min_width=256), align + bgr_image_warp Lanczos2",
               w=1920, h=1080, frames=240, seed=1, bits=8, clips=1, stabilizer=False),
    "c3": dict(name="4K single clip, 4-level pyramid (pyramid_min_width=256), align + bgr_image_warp Lanczos2",
               w=3840, h=2160, frames=120, seed=2, bits=8, clips=1, stabilizer=False),
    "c4": dict(name="batch of independent 1080p clips x 120 frames, 3-level pyramid, align + bgr_image_warp Lanczos2",
               w=1920, h=1080, frames=120, seed=1000, bits=8, clips=8, stabilizer=False),
    "c5": dict(name="4K 10-bit BGR clips x 60 frames, full stabilizer loop (L1 smoother, lag 10, Lanczos2 warp, crop 32)",
               w=3840, h=2160, frames=60, seed=2000, bits=10, clips=8, stabilizer=True),
}


def cpu_baseline(frames_host, params_kw, stabilizer, seconds_budget=20.0):
    """oracle (CPU restatement of the reference) on the host cores: one independent clip copy per thread, kernels
    single-threaded -- the reference's own multi-clip regime (grid_search_align.cpp:105-210)."""
    from oracle import oracle as O
    n = frames_host.shape[0]
    threads = max(1, min(os.cpu_count() or 1, 16))

    def run(count, k=None, done=None):
        if stabilizer:
            st = O.Stabilizer(warp_mode=O.WARP_LANCZOS2, **params_kw)
            for i in range(count):
                st.process(frames_host[i])
                if done is not None:
                    done[k] += 1
        else:
            a = O.Aligner(**params_kw)
            for i in range(count):
                ok, t = a.align_next(frames_host[i])
                O.bgr_image_warp(frames_host[i], t if ok else O.Transform.of())
                if done is not None:
                    done[k] += 1

    t0 = time.perf_counter()
    run(2)                                                     # calibrate on one thread to size the sample
    per_frame = (time.perf_counter() - t0) / 2
    sample = int(max(3, min(n, seconds_budget / max(per_frame, 1e-6))))
    done = [0] * threads
    th = [threading.Thread(target=run, args=(sample, k, done)) for k in range(threads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    what = "full stabilizer loop" if stabilizer else "align + Lanczos2 warp"
    # SURVEY 8(d) mode (i): ONE clip, the image-sized stages row-parallel over the same cores (the analogue of the .parallel(y)
    # of the reference's Halide schedules; the Gauss-Newton sums stay serial as sparse_ica.schedule.h has them)
    O.set_threads(threads)
    single_n = int(max(3, min(n, sample * threads // 4)))
    t0 = time.perf_counter()
    run(single_n)
    dt1 = time.perf_counter() - t0
    O.set_threads(1)
    return {"value": round(sum(done) / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "first %d frames of one clip, %s, %d threads x 1 clip copy each (the reference's many-clip regime, "
                      "grid_search_align.cpp:105-210), CPU restatement of the Halide path (not Halide)" % (sample, what, threads),
            "single_clip": {"value": round(single_n / dt1, 2), "unit": "frames/s", "threads": threads,
                            "sample": "first %d frames of one clip, stages row-parallel over %d threads" % (single_n, threads)}}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def spawn_ranks(n, argv, n_devices_hint=None):
    """Start ranks 0..n-1 of this script as child processes (fresh interpreters: nothing in this process has touched
    the GPU, and nothing is exec'ed from a process that has).  LOCAL_RANK i -> device i unless --device overrides it."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def roofline_4k(torch, capi, dev, stream, frames=32, reps=7):
    """bgr_image_warp where the north star quotes it: `frames` 4K u8 frames per launch, nothing else running."""
    W, H = 3840, 2160
    src = torch.randint(0, 256, (frames, H, W, 3), device=dev, dtype=torch.int32).to(torch.uint8)
    dst = torch.empty_like(src)
    ts = [capi.Transform.of(0.002, -0.0015, 3.3 + 0.37 * i, -2.7 - 0.21 * i) for i in range(frames)]
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_warp_pmc.json")))
    except Exception:
        pass
    out = {}
    for name, mode in (("exact", capi.WARP_LANCZOS2), ("fast", capi.WARP_LANCZOS2_FAST)):
        def run():
            capi.bgr_image_warp_batch_device(src.data_ptr(), frames, W, H, 3, 8, ts, dst.data_ptr(), mode, capi.BORDER_CLAMP,
                                             max_value=255, stream=stream.cuda_stream)
        run()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            run()
            b.record(stream)
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        ms.sort()
        med = ms[len(ms) // 2]
        nbytes = W * H * 3 * 2 * frames
        ach = nbytes / (med * 1e-3) / 1e9
        p = pmc.get(name, {})
        out[name] = {"kernel": "vs_k_bgr_warp_c3<u8,%s,clamp>" % ("lanczos2" if name == "exact" else "lanczos2 fast"),
                     "bound": "hbm", "binding": "valu", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": int(p["traffic_bytes_per_frame"] * frames) if "traffic_bytes_per_frame" in p else None,
                     "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                     "valu_instr_per_px": p.get("valu_instr_per_px"), "valu_frac": p.get("valu_frac"), "lds_frac": p.get("lds_frac"),
                     "counters": "valu_instr_per_px / valu_frac / lds_frac / traffic: rocprofv3 PMC passes of this kernel committed in "
                                 "profiles/r02_warp_pmc.json (not measured in this run); achieved: HIP events in this run"}
    del src, dst
    return out


def host_fed(torch, capi, dev, frames_dev, W, H, fmt, params_kw, reps=3):
    """Frames that start in (pageable) host memory: vs_aligner_align_batch(VS_MEM_HOST) cuts the batch into chunks and
    uploads chunk c+1 (own thread + stream) under the pipeline of chunk c.  Reported beside the link rate measured here."""
    import numpy as np
    host = frames_dev.cpu().numpy()
    if host.dtype == np.int16:
        host = host.view(np.uint16)
    n, nbytes = host.shape[0], host.nbytes
    pinned = torch.from_numpy(host.view(np.uint8).reshape(-1)).pin_memory()
    dbuf = torch.empty_like(pinned, device=dev)

    def best(fn):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    t_link = best(lambda: dbuf.copy_(pinned, non_blocking=True))
    del dbuf, pinned
    al = capi.Aligner(device=dev.index, **params_kw)
    res = {}

    def run():
        al.reset()
        res["r"] = al.align_batch(host)
    t = best(run)
    ref = capi.Aligner(device=dev.index, **params_kw)
    st_d, ts_d = ref.align_batch_device(frames_dev.data_ptr(), n, W, H, fmt)
    same = list(res["r"][0]) == list(st_d) and [x.tup() for x in res["r"][1]] == [x.tup() for x in ts_d]
    return {"value": round(n / t, 1), "unit": "frames/s", "input_GBps": round(nbytes / t / 1e9, 2),
            "pinned_h2d_GBps": round(nbytes / t_link / 1e9, 2), "of_pinned_h2d": round(t_link / t, 3), "frames": n,
            "identical_to_device_resident": same,
            "note": "alignment of a host-resident clip, PCIe-inclusive (never `value`); pipelined ingest, pageable host memory"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override the clip length")
    ap.add_argument("--clips-per-gpu", type=int, default=0)
    ap.add_argument("--select", default="device", choices=["host", "device"])
    ap.add_argument("--no-warp", action="store_true", help="alignment only (c2/c3/c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU run); gloo only to rehearse the N>1 flow on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="override LOCAL_RANK -> device (rehearsal on a 1-GPU box)")
    ap.add_argument("--default-levels", action="store_true",
                    help="reference default pyramid_min_width/height = 20 (6 levels at 1080p, 7 at 4K) instead of 256")
    ap.add_argument("--warp-mode", default="exact", choices=["exact", "fast"],
                    help="exact = VS_WARP_LANCZOS2 (bit-identical to the CPU restatement, the default and the parity claim); "
                         "fast = VS_WARP_LANCZOS2_FAST (opt-in fused-multiply-add variant, within 1 LSB)")
    ap.add_argument("--phase-correlate", action="store_true", help="aligner with phase_correlate = true (off in the reference's defaults)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (checks the RCCL path)")
    ap.add_argument("--no-roofline-4k", action="store_true", help="skip the isolated 32 x 4K bgr_image_warp measurement")
    ap.add_argument("--exclusive-solver", action="store_true", help="keep the solver kernel in VS_BATCH_EXCLUSIVE mode inside the overlapped step")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-resident (PCIe-inclusive) alignment measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be the launcher.  Nothing above has imported torch or made a HIP call.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    from video_stabilizer_amd import capi, synth
    from video_stabilizer_amd import dist as vsdist

    world, rank, local_rank = vsdist.env_world()
    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        # one process per GPU; "nccl" is RCCL on ROCm.  No data-path collective: clips are independent.
        dist = vsdist.init(args.dist_backend, rank, world, device_id=dev if args.dist_backend == "nccl" else None)
    red_dev = dev if args.dist_backend == "nccl" else None     # where the three report scalars are reduced

    wl = WORKLOADS[args.workload]
    W, H, bits = wl["w"], wl["h"], wl["bits"]
    n = args.frames or wl["frames"]
    n_clips = args.clips_per_gpu or wl["clips"]
    fmt = capi.FMT_BGR8 if bits == 8 else capi.FMT_BGR16
    max_value = 255 if bits == 8 else (1 << bits) - 1
    # global clip index of local clip j = rank + j*world (clip i -> rank i mod N); textures per rank, path per clip
    factory = synth.TorchClipFactory(W, H, wl["seed"] + 1000 * rank, dev, channels=3, bits=bits)
    # the rank's clips live back to back in one tensor, so the aligner can take all of them in one call
    all_frames = torch.empty((n_clips * n, H, W, 3), dtype=torch.uint8 if bits == 8 else torch.int16, device=dev)
    clips = [factory.make(n, wl["seed"] + 1000 * (rank + j * world), out=all_frames[j * n:(j + 1) * n])[0] for j in range(n_clips)]
    torch.cuda.synchronize()

    params_kw = {} if args.default_levels else dict(pyramid_min_width=256)
    if args.phase_correlate:
        params_kw["phase_correlate"] = 1
    stream = torch.cuda.current_stream()
    ev = []   # (start, end) events around the warp launches of the timed steps

    if wl["stabilizer"]:
        crop = 32
        stab = capi.Stabilizer(device=local_rank, warp_mode=capi.WARP_LANCZOS2, **params_kw)   # the library default is the reference's bilinear
        out_buf = torch.empty((n_clips * n, H - 2 * crop, W - 2 * crop, 3), dtype=clips[0].dtype, device=dev)
        aligner = None

        def step(timed):
            # all clips of the rank in one call (vs_stabilizer_process_clips): each clip through a fresh stabilizer,
            # alignment and warps of all clips batched together
            r, _ = stab.process_clips_device(all_frames.data_ptr(), n_clips, n, W, H, fmt, out_buf.data_ptr())
            return r
    else:
        aligner = capi.Aligner(device=local_rank,
                               select_mode=capi.SELECT_DEVICE if args.select == "device" else capi.SELECT_STL_HOST, **params_kw)
        warped = torch.empty_like(all_frames)
        N = n_clips * n
        # the timed step overlaps the warp of pass k with the alignment of pass k+1 (two streams): full batches go through the
        # small-footprint build of the solver kernel, which shares CUs with the warp grid (bit-identical results)
        if not args.no_warp and not args.exclusive_solver:
            aligner.set_batch_mode(capi.BATCH_SHARED)

        def step(timed, warp_mode=None):
            # all clips of the rank in one call (vs_aligner_align_clips): every stage is one launch over all clips
            status, ts = aligner.align_clips(N, n_clips, mem_ptr=all_frames.data_ptr(), w=W, h=H, fmt=fmt, raw=True)
            if not args.no_warp:
                if timed:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                if warp_mode is None:
                    warp_mode = capi.WARP_LANCZOS2 if args.warp_mode == "exact" else capi.WARP_LANCZOS2_FAST
                capi.bgr_image_warp_batch_device(all_frames.data_ptr(), N, W, H, 3, 8 if bits == 8 else 16, ts, warped.data_ptr(),
                                                 warp_mode, capi.BORDER_CLAMP, max_value=max_value, stream=stream.cuda_stream)
                if timed:
                    b.record(stream)
                    ev.append((a, b))
            return sum(status)

    def timed_loop(fn, k):
        """k calls of fn between barrier + synchronize on both sides; seconds on this rank, last return value"""
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            r = fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return time.perf_counter() - t0, r

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if aligner:
        aligner.enable_timing(True)
    dt, good = timed_loop(lambda: step(True), args.steps)
    # whole-job numbers: max seconds over ranks, frames summed over ranks (the only collectives of the run)
    dt, total_frames, total_good = vsdist.aggregate(dt, n * n_clips * args.steps, int(good) * args.steps, device=red_dev)
    tm = aligner.timings() if aligner else None

    align_only = None
    if aligner and not args.no_warp:
        # second, separately reported figure: the alignment stages alone (BASELINE configs[1] read literally)
        def fn():
            aligner.align_clips(N, n_clips, mem_ptr=all_frames.data_ptr(), w=W, h=H, fmt=fmt, raw=True)
        aligner.set_batch_mode(capi.BATCH_EXCLUSIVE)          # nothing else runs: one 512-thread workgroup per pair
        aligner.enable_timing(True)
        dt_a, _ = timed_loop(fn, args.steps)
        dt_a, frames_a, _ = vsdist.aggregate(dt_a, n * n_clips * args.steps, 0, device=red_dev)
        align_only = (dt_a, frames_a, aligner.timings())

    fast_warp = None
    if aligner and not args.no_warp and not args.exclusive_solver:
        aligner.set_batch_mode(capi.BATCH_SHARED)
    if aligner and not args.no_warp and args.warp_mode == "exact":
        # third figure: the same step with the tolerance-gated warp arithmetic (VS_WARP_LANCZOS2_FAST: within the north star's
        # "1 ULP of the Lanczos path", tests/test_warp_fast_gpu.py) -- reported beside `value`, never as `value`
        step(False, capi.WARP_LANCZOS2_FAST)
        dt_f, _ = timed_loop(lambda: step(False, capi.WARP_LANCZOS2_FAST), args.steps)
        dt_f, frames_f, _ = vsdist.aggregate(dt_f, n * n_clips * args.steps, 0, device=red_dev)
        fast_warp = (dt_f, frames_f)

    if rank == 0:
        def stage_table(t):
            return {k: {"ms_per_step": round(v["ms"] / args.steps, 4), "launches_per_step": v["launches"] // max(1, args.steps)}
                    for k, v in t.items() if isinstance(v, dict) and v["launches"]}
        out = {
            "metric": "aligned frames/sec", "value": round(total_frames / dt, 2), "unit": "frames/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "dist_backend": (args.dist_backend if dist is not None else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8" if bits == 8 else "u16", "data": "synthetic",
            "config": {"workload": wl["name"] if not args.default_levels else
                       wl["name"].split(",")[0] + ", reference default pyramid (pyramid_min_width = pyramid_min_height = 20)",
                       "frames_per_clip": n, "clips_per_gpu": n_clips, "width": W, "height": H,
                       "bits": bits, "clip_seeds": "clip i -> rank i mod N; path seed %d + 1000 i" % wl["seed"],
                       "selection": "std::nth_element on the host" if args.select == "host"
                       else "on-device replica of libstdc++ nth_element (same survivors, same order)",
                       "phase_correlate": bool(args.phase_correlate),
                       "warp": None if (args.no_warp and not wl["stabilizer"]) else
                       ("bgr_image_warp lanczos2" if args.warp_mode == "exact" else "bgr_image_warp lanczos2, opt-in fast mode (<= 1 LSB from exact)"), "resident": "HBM"},
            ("outputs_per_step" if wl["stabilizer"] else "aligned_per_step"): total_good // args.steps,
        }
        if tm:
            out["stages"] = stage_table(tm)
            out["gn_iterations_per_frame"] = round(tm["gn_iterations"] / max(1, tm["frames"]), 2)
        if align_only:
            dt_a, frames_a, tm_a = align_only
            out["align_only"] = {"value": round(frames_a / dt_a, 2), "unit": "frames/s",
                                 "ms_per_step": round(1e3 * dt_a / args.steps, 4), "stages": stage_table(tm_a),
                                 "note": "same clips, alignment stages only (no warp launch competing for the CUs)"}
        if fast_warp:
            out["fast_warp"] = {"value": round(fast_warp[1] / fast_warp[0], 2), "unit": "frames/s",
                                "ms_per_step": round(1e3 * fast_warp[0] / args.steps, 4),
                                "note": "same step with bgr_image_warp in VS_WARP_LANCZOS2_FAST (fused multiply-adds; float output within the "
                                        "ULP bound and integer output <= 1 LSB / >= 99.99 % identical, tests/test_warp_fast_gpu.py)"}
        if ev:
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)          # one launch per step over all the rank's frames
            bytes_per_launch = W * H * 3 * 2 * (1 if bits == 8 else 2) * n * n_clips  # W*H*3*(in+out) B per frame
            achieved = bytes_per_launch / (ms * 1e-3) / 1e9
            # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json), scaled to
            # this launch's frame count; null when there is no profile for this frame format
            traffic = None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))
                key, per = {(1920, 8): ("c2_1080p_240_frames", 240), (3840, 8): ("c3_4k_32_frames", 32)}[(W, bits)]
                traffic = int(tj[key]["traffic_bytes"] / per * n * n_clips)
            except Exception:
                pass
            out["roofline"] = {"kernel": "vs_k_bgr_warp_c3<lanczos2,clamp> (bgr_image_warp)", "bound": "hbm", "binding": "valu",
                               "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                               "traffic_source": "scaled from the committed PMC passes (profiles/r02_traffic.json), not measured in this run",
                               "launch_ms": round(ms, 4), "bytes_per_launch": bytes_per_launch,
                               "note": "VALU-issue-bound, not HBM-bound: ~230 VALU instructions per output pixel in the "
                                       "reference's exact fp32 order (DESIGN.md, profiles/r02_bgr_image_warp.md); launches "
                                       "overlap the next clip's aligner kernels"}
        if not args.no_roofline_4k:
            out["roofline_4k"] = roofline_4k(torch, capi, dev, stream)
        if not args.no_host_fed and aligner is not None:
            out["host_fed"] = host_fed(torch, capi, dev, clips[0], W, H, fmt, params_kw)
        if not args.no_cpu_baseline and world == 1:
            fh = clips[0][: min(n, 64)].cpu().numpy()
            if bits != 8:
                fh = fh.view("uint16")
            out["cpu_baseline"] = cpu_baseline(fh, params_kw, wl["stabilizer"])
            out["cpu_baseline"]["cpu_model"] = cpu_model()
            out["cpu_baseline"]["hardware_threads"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
