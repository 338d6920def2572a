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
value = ALIGNED frames per second, whole job (all ranks): frames for which AlignNextFrame returns true (the first frame of a
clip has no predecessor: 239 of 240 count).  One process per GPU; clips are independent, so ranks share nothing but the
barriers that bracket the timed region and one max/sum all-reduce for the report ("weak" scaling).
The warp of the timed step is bgr_image_warp in its SEPARABLE form (VS_WARP_LANCZOS2_SEP, `value_warp_mode`: the contracted weights
and taps, summed rows first, then columns, over the product of the two 1-D weight sums -- equal to generators.cpp:687-697 in real
arithmetic, a reassociation inside the slack of the reference's own non-strict_float build), admitted as `value` by SURVEY 8(d)'s
integer gate, which every run re-checks against the UN-contracted oracle (`parity.separable_vs_exact`: max 1 LSB, >= 99.99 % of the
samples identical; a broken gate fails the run).  The same step with the un-contracted sampler is `exact_warp` (the figure to compare
across rounds: rounds 1-3 reported it as `value`), with the contracted one (round 4's `value`) `contracted_warp`.
Timing: after the driver's --warmup steps an UNTIMED pre-roll keeps stepping until >= 150 ms of launches have run (the shader clock
needs ~40 ms of continuous work to settle: tools/clock_settling.py); then the driver's loop -- exactly --steps steps between barrier +
synchronize -- runs three times back to back: `value` / `ms_per_step` are the MEDIAN repeat, `value_spread` has min / median / max,
`shader_clock_mhz` the clock a probe kernel issued directly behind the loops sees (s_memtime / s_memrealtime).

The JSON line also carries
  roofline      the dominant kernel of the timed region (bgr_image_warp): algorithmic bytes / mean launch time, HIP
                events on the launch stream, against the 8 TB/s HBM peak; `traffic` scaled from the committed PMC passes
  default_warp  (in the headline's object and in c3) the reference's OWN pipeline on the workload: align + cv::warpAffine's fixed-point bilinear
                (VS_WARP_BILINEAR_CV), constant border, the measured transform as the forward map (stabilizer.cpp:97-99); both solver modes,
                its own in-step roofline (+ live traffic) and stage table; never `value`
  roofline.at_4k / roofline_4k_summary
                the `value` warp mode at 4K inside the headline's roofline object; {mode: [us per 4K frame, fraction of 8 TB/s]} as the LAST
                key of the line (a record that keeps only the tail still shows the north star's operating point)
  ranks         (N > 1) the roll call made before anything is timed: every rank's device, backend and clip split; two ranks on one
                device while the node shows a device for each end the job (exit code 5); "scaling_curve_measured": false
  roofline_4k   the same kernel where the north star quotes it: 32 x 4K frames per launch, isolated, after the timed
                region (exact and contracted arithmetic), with the VALU / LDS busy fractions of the committed PMC passes --
                the kernel is VALU-issue-bound, the HBM fraction is what that leaves
  c3            (default run, one GPU) the 4K half of the metric: BASELINE configs[2] through the same step, a few steps
  c5            (default run, one GPU) BASELINE configs[4]: 8 clips x 60 frames 4K 10-bit through the full stabilizer loop
  drop_in       the reference's own call pattern (video_test.cpp:106, stabilizer.cpp:19): HOST frames, one vs_stabilizer_process /
                one vs_aligner_align_next per frame, 1080p and 4K -- frames/s and ms per call, PCIe-inclusive, never `value` --
                beside the oracle making the same calls
  c4_strong     (default run) BASELINE configs[3] as a strong-scaling leg: 64 clips in total, 64 / N per rank, per-rank seconds
  parity        the gate SURVEY 8(d) asks for with every benchmark: the GPU results for the first frames of the clip against
                the CPU restatement that cpu_baseline runs anyway; a broken gate makes the run fail (exit code 3)
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
# dense fp32 vector peak: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = one wave-instruction per 2 cycles per SIMD (MI355X_MICROARCH.md)
VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2

# the three members of the Lanczos2 sampler family: bench name -> (capi constant, oracle twin constant, kernel label)
WARP_MODES = {
    "separable": ("WARP_LANCZOS2_SEP", "WARP_LANCZOS2_SEPARABLE", "lanczos2 separable (VS_WARP_LANCZOS2_SEP)"),
    "contracted": ("WARP_LANCZOS2_FAST", "WARP_LANCZOS2_CONTRACTED", "lanczos2 contracted (VS_WARP_LANCZOS2_FAST)"),
    "exact": ("WARP_LANCZOS2", "WARP_LANCZOS2", "lanczos2"),
}
WARP_BENCH_NAME = {"separable": "sep", "contracted": "fast", "exact": "lanczos2", "bilinear_cv": "cv"}          # tools/warp_bench.py --mode
PREROLL_SECONDS = 0.15
REPEATS = 3

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


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def usable_threads():
    """hardware threads this process can keep busy (affinity mask, capped by the container's CPU quota), and the host's total"""
    total = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = total
    q = cgroup_cpu_quota()
    if q is not None:
        avail = max(1, min(avail, int(q + 0.999)))
    return avail, total


def cpu_baseline(frames_host, params_kw, stabilizer, seconds_budget=20.0, record=None, select_rule=0):
    """oracle (CPU restatement of the reference) on the host cores: one independent clip copy per thread, kernels
    single-threaded -- the reference's own multi-clip regime (grid_search_align.cpp:105-210).  `record` (a list) receives
    thread 0's per-frame results (ok, transform, iterations, fail_reason): the parity gate reads them, no extra CPU work."""
    from oracle import oracle as O
    n = frames_host.shape[0]
    avail, total = usable_threads()
    threads = max(1, min(avail, 16))

    def run(count, k=None, done=None, rec=None):
        if stabilizer:
            st = O.Stabilizer(select_rule=select_rule, warp_mode=O.WARP_LANCZOS2, **params_kw)
            for i in range(count):
                st.process(frames_host[i])
                if done is not None:
                    done[k] += 1
        else:
            a = O.Aligner(select_rule=select_rule, **params_kw)
            for i in range(count):
                ok, t = a.align_next(frames_host[i])
                if rec is not None:
                    d = a.debug()
                    rec.append((ok, t.tup(), list(d.iterations[:d.levels]), int(d.fail_reason)))
                O.bgr_image_warp(frames_host[i], t if ok else O.Transform.of())
                if done is not None:
                    done[k] += 1

    def many_clip(nthreads, sample, rec=None):
        done = [0] * nthreads
        th = [threading.Thread(target=run, args=(sample, k, done, rec if k == 0 else None)) for k in range(nthreads)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return sum(done) / (time.perf_counter() - t0)

    t0 = time.perf_counter()
    run(2)                                                     # calibrate on one thread to size the sample
    per_frame = (time.perf_counter() - t0) / 2
    sample = int(max(3, min(n, seconds_budget / max(per_frame, 1e-6))))
    rate16 = many_clip(threads, sample, record)
    what = "full stabilizer loop" if stabilizer else "align + Lanczos2 warp"
    out = {"value": round(rate16, 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "first %d frames of one clip, %s, %d threads x 1 clip copy each (the reference's many-clip regime, "
                     "grid_search_align.cpp:105-210), CPU restatement of the Halide path (not Halide)" % (sample, what, threads)}
    # "over all host cores" -- every hardware thread this process may run on, same regime, a sample sized so that
    # the leg takes ~10 s whatever the thread count turns out to be worth (a 2-frame probe per thread sizes it)
    if avail > threads:
        probe = many_clip(avail, 2)
        sample_all = int(max(3, min(n, 10.0 * probe / avail)))
        rate_all = many_clip(avail, sample_all)
        out["all_cores"] = {"value": round(rate_all, 2), "unit": "frames/s", "cores": avail,
                            "sample": "first %d frames of one clip per thread, %d threads" % (sample_all, avail)}
    else:
        out["all_cores"] = {"value": round(rate16, 2), "unit": "frames/s", "cores": threads,
                            "sample": "the process may run on %d hardware threads: the figure above already uses all of them" % avail}
    # SURVEY 8(d) mode (i): ONE clip, the image-sized stages row-parallel over the same cores (the analogue of the .parallel(y)
    # of the reference's Halide schedules; the Gauss-Newton sums stay serial as sparse_ica.schedule.h has them)
    O.set_threads(threads)
    single_n = int(max(3, min(n, sample * threads // 4)))
    t0 = time.perf_counter()
    run(single_n)
    dt1 = time.perf_counter() - t0
    O.set_threads(1)
    out["single_clip"] = {"value": round(single_n / dt1, 2), "unit": "frames/s", "threads": threads,
                          "sample": "first %d frames of one clip, stages row-parallel over %d threads" % (single_n, threads)}
    out["hardware_threads"] = total
    out["usable_threads"] = avail
    out["cpu_quota"] = cgroup_cpu_quota()
    return out


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
    # rank 0's stdout (the one JSON line) is drained on a thread, so that the loop below can watch EVERY rank: the first rank that ends with
    # an error ends the job -- the others are sitting in a barrier that will never complete -- and its exit code is the job's (what torchrun
    # does for the driver's launches).  Only the exact children started above are ever signalled.
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, live = 0, set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    sys.stderr.write("bench.py: rank %d ended with exit code %d: stopping the other ranks\n" % (r, code))
                    break
        if live and rc == 0:
            time.sleep(0.05)
    for r in sorted(live):
        procs[r].terminate()
    for r in sorted(live):
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    reader.join(timeout=20)
    if rc == 0:
        sys.stdout.write(b"".join(chunks).decode())
        sys.stdout.flush()
    return rc


def load_profile(*names):
    for nm in names:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", nm))), nm
        except Exception:
            pass
    return {}, None


def measure_traffic_live(W, H, n_frames, bits, timeout_s=150, mode="separable"):
    """HBM-side bytes of ONE warp launch of the headline's shape, measured in THIS run: two rocprofv3 counter passes (FETCH_SIZE and
    WRITE_SIZE on their own: they do not fit one pass on gfx950) of tools/warp_bench.py as child processes -- `--pmc` alone, no trace
    domain, the program itself behind `--`, TMPDIR=/tmp, as MI355X_MICROARCH.md prescribes.  FETCH_SIZE is doubled (gfx950 tallies
    128-byte read requests at 64 bytes; re-checked on this kernel's 12-byte-per-lane loads in round 1: 0.500x), WRITE_SIZE is exact;
    counter unit KiB.  -> (bytes per launch, description) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process is itself running under a profiler (no nested counter passes)"
    tmp = tempfile.mkdtemp(prefix="vs_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    t0 = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            left = timeout_s - (time.perf_counter() - t0)
            if left < 20:
                return None, "the counter passes ran out of their %d s budget" % timeout_s
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "warp_bench.py"),
                   "--mode", WARP_BENCH_NAME[mode], "--w", str(W), "--h", str(H), "--frames", str(n_frames), "--reps", "2", "--bits", "8" if bits == 8 else "16"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=left, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            vals, ids = 0.0, set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "bgr_warp_c" in row["Kernel_Name"] and row["Counter_Name"] == ctr:       # (vs_k_bgr_warp_c3 and vs_k_bgr_warp_cv_c3[_u16])
                        vals += float(row["Counter_Value"])
                        ids.add(row["Dispatch_Id"])
            if not ids:
                return None, "rocprofv3 --pmc %s gave no rows for the warp kernel (exit code %d)" % (ctr, r.returncode)
            got[ctr] = vals / len(ids)
    except Exception as e:                                   # noqa: BLE001 -- a profiler problem must not cost the line its headline
        return None, "counter pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return int((2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024), (
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the same launch shape (tools/warp_bench.py, "
        "%d x %dx%d frames per launch, %s form), FETCH_SIZE doubled per the gfx950 correction, %.0f s" % (n_frames, W, H, mode, time.perf_counter() - t0))


def roofline_4k(torch, capi, dev, stream, frames=32, reps=7):
    """bgr_image_warp where the north star quotes it: `frames` 4K u8 frames per launch, nothing else running."""
    W, H = 3840, 2160
    src8 = torch.randint(0, 256, (frames, H, W, 3), device=dev, dtype=torch.int32).to(torch.uint8)
    dst8 = torch.empty_like(src8)
    ts = [capi.Transform.of(0.002, -0.0015, 3.3 + 0.37 * i, -2.7 - 0.21 * i) for i in range(frames)]
    pmc, pmc_name = load_profile("r05_warp_pmc.json", "r04_warp_pmc.json", "r03_warp_pmc.json")
    out = {}
    for name, mode, key in (("exact", capi.WARP_LANCZOS2, "exact"), ("contracted", capi.WARP_LANCZOS2_FAST, "contracted"),
                            ("separable", capi.WARP_LANCZOS2_SEP, "separable"),
                            ("bilinear_cv", capi.WARP_BILINEAR_CV, "bilinear_cv"),
                            ("bilinear", capi.WARP_BILINEAR, "bilinear"), ("bilinear_10bit", capi.WARP_BILINEAR, "bilinear_10bit"),
                            ("bilinear_cv_10bit", capi.WARP_BILINEAR_CV, "bilinear_cv_10bit")):
        bits = 16 if name.endswith("_10bit") else 8
        if bits == 16 and src8 is not None:                  # 10-bit frames in 16-bit containers (configs[4]'s format): twice the bytes per pixel
            src8 = dst8 = None
            src = torch.randint(0, 1024, (frames, H, W, 3), device=dev, dtype=torch.int32).to(torch.int16)
            dst = torch.empty_like(src)
        elif bits == 8:
            src, dst = src8, dst8

        def run():
            capi.bgr_image_warp_batch_device(src.data_ptr(), frames, W, H, 3, bits, ts, dst.data_ptr(), mode, capi.BORDER_CLAMP,
                                             max_value=255 if bits == 8 else 1023, stream=stream.cuda_stream)
        # The card's shader clock takes ~40 ms of continuous work to settle (tools/clock_settling.py: 64 -> 56 -> 49 us per frame over the
        # first 5 / 12 / 40 ms after an idle spell, flat from there on): launches back to back for >= 80 ms first, then `reps` more,
        # still back to back, each between two events on the launch stream.
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.08:
            for _ in range(4):
                run()
            torch.cuda.synchronize()
        evs = []
        for _ in range(4 + reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            run()
            b.record(stream)
            evs.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs[4:])
        med = ms[len(ms) // 2]
        nbytes = W * H * 3 * 2 * frames * (bits // 8)
        ach = nbytes / (med * 1e-3) / 1e9
        p = pmc.get(key, {})
        ipp = p.get("valu_instr_per_px")
        # VALU-peak fraction: wave-instructions the launch issues / what 1024 SIMDs issue in that time at one per 2 cycles
        valu_peak_frac = round(ipp * W * H * frames / 64.0 / (med * 1e-3) / VALU_WAVE_INSTR_PER_S, 4) if ipp else None
        if name == "bilinear_10bit":
            out[name] = {"kernel": "vs_k_bgr_warp_c3<u16,bilinear,clamp> (word tile)", "bound": "hbm", "binding": "hbm + valu",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                         "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                         "parity": "np.array_equal with the CPU restatement (VSO_WARP_BILINEAR, 10-bit)"}
            continue
        if name == "bilinear_cv_10bit":
            out[name] = {"kernel": "vs_k_cv_tables + vs_k_bgr_warp_cv_c3_u16<clamp> (word tile, v_dot2_u32_u16 taps while the samples stay below 2^14)", "bound": "hbm",
                         "binding": "hbm + valu", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                         "traffic": None, "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                         "parity": "np.array_equal with the CPU restatement (VSO_WARP_BILINEAR_CV on 16-bit containers: OpenCV's float-weight form)"}
            continue
        if name == "bilinear_cv":
            # the stabilizer's DEFAULT sampler = the reference's own warp (cv::warpAffine INTER_LINEAR, stabilizer.cpp:97-99 -> imgproc.cpp:472):
            # OpenCV's fixed-point bilinear restated, integer work end to end; vector issue and the memory side together (profiles/r05_warp_cv.md)
            pc, _ = load_profile("r06_pmc_bilinear_cv.json", "r05_pmc_bilinear_cv.json")
            out[name] = {"kernel": "vs_k_cv_tables + vs_k_bgr_warp_cv_c3<clamp> (OpenCV's coordinate tables once per frame; byte tile, v_dot2_u32_u16 taps)", "bound": "hbm",
                         "binding": "valu issue (~34 integer-class vector instructions per 64 pixels, ~80 % of a SIMD's issue time) and the memory side (56 MB per 4K frame through "
                         "L2 with the tile halo) together; the events bracket the CALL (table kernel + dependent-launch gap + warp kernel): the warp kernel alone is ~3 % "
                         "faster (profiles/r06_roofline4k.md, r06_warp_cv.md)", "valu_instr_per_px": pc.get("valu_instr_per_px"),
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                         "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                         "parity": "np.array_equal with the CPU restatement (VSO_WARP_BILINEAR_CV: OpenCV 4.x's published fixed-point path, parity "
                                   "unpinned (OpenCV version)); transforms = forward maps, as warpBySimilarityTransform hands them to cv::warpAffine"}
            continue
        if name == "bilinear":
            # the Halide sampler's float lerp (generators.cpp:148-163; the stabilizer's default until round 5): 88 vector instructions per
            # pixel (counted: profiles/r04_pmc_bilinear.json), VALU-issue-bound like the Lanczos kernels (profiles/r04_ab_warp_bilinear.md)
            out[name] = {"kernel": "vs_k_bgr_warp_c3<u8,bilinear,clamp> (byte tile)", "bound": "hbm", "binding": "valu", "valu_instr_per_px_r04": 87.8,
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                         "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                         "parity": "np.array_equal with the CPU restatement (VSO_WARP_BILINEAR)"}
            continue
        out[name] = {"kernel": "vs_k_bgr_warp_c3<u8,%s,clamp>" % WARP_MODES[name][2],
                     "bound": "hbm", "binding": "valu", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": int(p["traffic_bytes_per_frame"] * frames) if "traffic_bytes_per_frame" in p else None,
                     "us_per_frame": round(1e3 * med / frames, 2), "frames_per_launch": frames, "bytes_per_launch": nbytes,
                     "valu_instr_per_px": ipp, "valu_peak_frac": valu_peak_frac, "valu_frac": p.get("valu_frac"), "lds_frac": p.get("lds_frac"),
                     "parity": "np.array_equal with the CPU restatement (VSO_%s)" % WARP_MODES[name][1],
                     "counters": "valu_instr_per_px / valu_frac / lds_frac / traffic: rocprofv3 PMC passes of this kernel committed in "
                                 "profiles/%s (not measured in this run); achieved, valu_peak_frac: HIP events in this run" % pmc_name}
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


def drop_in(capi, dev_index, frames_host, params_kw, calls=48, cpu_budget_s=6.0):
    """The reference's own call pattern: every frame arrives in (pageable) host memory and is handed to ONE
    VideoStabilizer::processFrame (video_test.cpp:106) / ONE VideoAligner::AlignNextFrame (stabilizer.cpp:19) call, which
    returns the result to host memory before the next frame is offered.  No batching, no frame look-ahead: upload, compute and
    download of a frame are serial by the API's contract.  PCIe-inclusive, never `value`.  The oracle (CPU restatement, one
    thread: the reference's per-call path has no frame-level parallelism either) makes the same calls on a bounded sample."""
    from oracle import oracle as O
    n = min(calls, frames_host.shape[0])
    H, W = frames_host.shape[1:3]
    res = {"frames": "%dx%d %s, host memory (pageable)" % (W, H, "u8" if frames_host.dtype.itemsize == 1 else "u16 (10-bit)"), "calls": n}

    def timed(make, call, warm):
        h = make()
        for f in frames_host[:warm]:
            call(h, f)
        h.reset()                                      # a new clip on warm buffers: the timed calls allocate nothing
        t0 = time.perf_counter()
        got = sum(1 for f in frames_host[:n] if call(h, f))
        dt = time.perf_counter() - t0
        return {"frames_per_s": round(n / dt, 1), "ms_per_call": round(1e3 * dt / n, 4), "calls": n, "results": got}

    res["align_next"] = timed(lambda: capi.Aligner(device=dev_index, **params_kw), lambda h, f: h.align_next(f)[0], 3)
    res["align_next"]["what"] = "vs_aligner_align_next per frame (alignment.hpp:55-58); results = frames aligned"
    # library defaults = the reference's processFrame: cv::warpAffine's bilinear, black border, crop 32, lag 10 (stabilizer.hpp:13-30)
    res["process_frame"] = timed(lambda: capi.Stabilizer(device=dev_index, **params_kw), lambda h, f: h.process(f) is not None, 14)
    res["process_frame"]["what"] = ("vs_stabilizer_process per frame, library defaults (VS_WARP_BILINEAR_CV: cv::warpAffine's fixed-point INTER_LINEAR as "
                                    "stabilizer.cpp:97-99 calls it); results = frames returned")
    res["process_frame_lanczos2"] = timed(lambda: capi.Stabilizer(device=dev_index, warp_mode=capi.WARP_LANCZOS2_FAST, warp_border=capi.BORDER_CLAMP, **params_kw),
                                          lambda h, f: h.process(f) is not None, 14)
    res["process_frame_lanczos2"]["what"] = "the same with bgr_image_warp Lanczos2 (contracted), clamp border"

    def cpu_timed(make, call, count):
        h = make()
        t0 = time.perf_counter()
        k = 0
        for f in frames_host[:count]:
            call(h, f)
            k += 1
            if time.perf_counter() - t0 > cpu_budget_s and k >= 3:
                break
        dt = time.perf_counter() - t0
        return {"frames_per_s": round(k / dt, 2), "ms_per_call": round(1e3 * dt / k, 2), "calls": k, "threads": 1}

    res["cpu_align_next"] = cpu_timed(lambda: O.Aligner(**params_kw), lambda h, f: h.align_next(f), n)
    res["cpu_process_frame"] = cpu_timed(lambda: O.Stabilizer(**params_kw), lambda h, f: h.process(f), n)
    res["note"] = ("one call per frame, result on the host before the next call: the API's serial contract (H2D + every kernel + D2H in "
                   "sequence); cpu_* = the oracle (kind `port`, parity unpinned) making the same calls on one thread, bounded to ~%g s each" % cpu_budget_s)
    return res


class AlignWarp:
    """The timed step for the align + warp workloads (c2 / c3 / c4): the rank's clips resident in HBM, one aligner handle, one
    output buffer.  The warp of pass k (caller's stream) overlaps the alignment of pass k+1 (the handle's stream)."""

    def __init__(self, torch, capi, synth, dev, wl, n, n_clips, seeds, params_kw, args, factory_seed):
        self.torch, self.capi, self.dev, self.wl, self.n, self.n_clips, self.args = torch, capi, dev, wl, n, n_clips, args
        self.W, self.H, self.bits = wl["w"], wl["h"], wl["bits"]
        self.fmt = capi.FMT_BGR8 if self.bits == 8 else capi.FMT_BGR10
        self.max_value = 255 if self.bits == 8 else (1 << self.bits) - 1
        factory = synth.TorchClipFactory(self.W, self.H, factory_seed, dev, channels=3, bits=self.bits)
        dt = torch.uint8 if self.bits == 8 else torch.int16
        # the rank's clips live back to back in one tensor, so the aligner can take all of them in one call
        self.frames = torch.empty((n_clips * n, self.H, self.W, 3), dtype=dt, device=dev)
        self.clips = [factory.make(n, seeds[j], out=self.frames[j * n:(j + 1) * n])[0] for j in range(n_clips)]
        torch.cuda.synchronize()
        self.N = n_clips * n
        self.stream = torch.cuda.current_stream()
        self.aligner = capi.Aligner(device=dev.index,
                                    select_mode={"device": capi.SELECT_DEVICE, "stable": capi.SELECT_STABLE, "host": capi.SELECT_STL_HOST}[args.select],
                                    **params_kw)
        self.warped = None if args.no_warp else torch.empty_like(self.frames)
        self.ev = []
        self.shared(True)

    def shared(self, on):
        # the timed step overlaps the warp of pass k with the alignment of pass k+1 (two streams): full batches then go through
        # the small-footprint build of the solver kernel, which shares CUs with the warp grid (bit-identical results)
        on = on and not self.args.no_warp and not self.args.exclusive_solver
        self.aligner.set_batch_mode(self.capi.BATCH_SHARED if on else self.capi.BATCH_EXCLUSIVE)

    def align(self):
        return self.aligner.align_clips(self.N, self.n_clips, mem_ptr=self.frames.data_ptr(), w=self.W, h=self.H, fmt=self.fmt, raw=True)

    def warp(self, ts, mode, timed=False, border=None):
        torch = self.torch
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
        self.capi.bgr_image_warp_batch_device(self.frames.data_ptr(), self.N, self.W, self.H, 3, 8 if self.bits == 8 else 16, ts,
                                              self.warped.data_ptr(), mode, self.capi.BORDER_CLAMP if border is None else border, max_value=self.max_value,
                                              stream=self.stream.cuda_stream)
        if timed:
            b.record(self.stream)
            self.ev.append((a, b))

    def step(self, timed=False, mode=None, border=None):
        status, ts = self.align()
        if not self.args.no_warp:
            if mode is None:
                mode = getattr(self.capi, WARP_MODES[self.args.warp_mode][0])
            self.warp(ts, mode, timed, border)
        return sum(status)

    def default_warp_leg(self, timed_loop, steps, aggregate=None):
        """The reference's own pipeline on this workload: AlignNextFrame + warpBySimilarityTransform = cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT) with the
        measured transform as the FORWARD map (stabilizer.cpp:97-99 -> imgproc.cpp:472-481) -- VS_WARP_BILINEAR_CV, the library's default stabilizer warp.
        That warp takes a quarter of the alignment pass's time, so the step is alignment-bound: measured with the solver kernel in both batch modes
        (sharing CUs with the previous pass's warp, as `value` runs it, and exclusive); the faster one leads.  Its own in-step roofline (HIP events on the
        launch stream around every warp launch of the timed loop) and the aligner's stage table ride along."""
        capi, torch = self.capi, self.torch
        saved_ev = self.ev
        modes = {}
        for solver in ("shared", "exclusive"):
            self.shared(solver == "shared")
            self.step(False, capi.WARP_BILINEAR_CV, capi.BORDER_CONSTANT)
            torch.cuda.synchronize()
            self.ev = []
            self.aligner.enable_timing(True)
            dt, good = timed_loop(lambda: self.step(True, capi.WARP_BILINEAR_CV, capi.BORDER_CONSTANT), steps)
            tm = self.aligner.timings()
            if aggregate is not None:
                dt, _, good_total = aggregate(dt, self.N * steps, int(good) * steps)
            else:
                good_total = int(good) * steps
            launch_ms = sum(a.elapsed_time(b) for a, b in self.ev) / max(1, len(self.ev))
            modes[solver] = dict(dt=dt, good=good_total, tm=tm, launch_ms=launch_ms)
        self.ev = saved_ev
        self.shared(True)
        best = min(modes, key=lambda k: modes[k]["dt"])
        m = modes[best]
        nbytes = self.W * self.H * 3 * 2 * (1 if self.bits == 8 else 2) * self.N
        ach = nbytes / (m["launch_ms"] * 1e-3) / 1e9
        return {"value": round(m["good"] / m["dt"], 2), "unit": "frames/s", "ms_per_step": round(1e3 * m["dt"] / steps, 4), "solver": best,
                "by_solver_mode": {k: round(v["good"] / v["dt"], 2) for k, v in modes.items()},
                "warp": "VS_WARP_BILINEAR_CV (cv::warpAffine INTER_LINEAR restated: OpenCV's fixed-point bilinear), VS_BORDER_CONSTANT, the measured transform as the "
                        "forward map: what stabilizer.cpp:97-99 -> imgproc.cpp:472-481 does with every frame",
                "roofline": {"kernel": "vs_k_bgr_warp_cv_c3<constant> (byte tile, v_dot2_u32_u16 taps)" if self.bits == 8 else "vs_k_bgr_warp_cv_c3_u16<constant>",
                             "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                             "launch_ms": round(m["launch_ms"], 4), "bytes_per_launch": nbytes,
                             "note": "in-step: the launch runs beside the next pass's aligner kernels (HIP events on the launch stream, mean over the timed loop)"},
                "stages": stage_table(m["tm"], steps),
                "gn_iterations_per_frame": round(m["tm"]["gn_iterations"] / max(1, m["tm"]["frames"]), 2),
                "note": "never `value` (not a Lanczos2 bgr_image_warp): the figure a drop-in user of the reference's pipeline gets on this workload, frames resident in HBM"}

    def free(self):
        self.frames = self.warped = self.clips = None
        self.aligner = None
        self.torch.cuda.empty_cache()


def stage_table(t, steps):
    return {k: {"ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": v["launches"] // max(1, steps)}
            for k, v in t.items() if isinstance(v, dict) and v["launches"]}


def mode_vs_exact_gate(G, pairs, mode="contracted"):
    """SURVEY 8(d) integer gate over (GPU frame of `mode`, un-contracted oracle frame) pairs: max |d| <= 1 LSB and the
    fraction of identical samples >= 0.9999 on every frame"""
    mx, least, ok = 0, 1.0, True
    for got, want in pairs:
        good, info = G.integer_gate(got, want)
        ok = ok and good
        mx = max(mx, info["max_abs_diff_lsb"])
        least = min(least, info["identical_fraction"])
    return {"pass": bool(ok), "max_abs_diff_lsb": mx, "least_identical_fraction": round(least, 7),
            "gate": "max |d| <= 1 LSB and identical fraction >= %.4f per frame (SURVEY 8(d), integer modes)" % G.INTEGER_IDENTICAL_MIN,
            "gpu": "VS_%s (%s)" % (WARP_MODES[mode][0], mode), "cpu": "VSO_WARP_LANCZOS2 (un-contracted: the reference's written order)",
            "float_mode": "not part of `value` (integer outputs only): the formula of SURVEY 8(d) for float output is missed by every "
                          "evaluation order, the reference's own included (tests/test_warp_gate_cpu.py records by how much)"}


def parity_gate(torch, capi, aw, params_kw, oracle_rec, frames_host, warp_frames=(1, 2, 17, 40)):
    """"parity gates run with every benchmark".  The GPU side is the path the timed step runs (full batch, device
    selection, exact and contracted warps of the whole clip); the CPU side is what cpu_baseline's first thread computed on the
    first frames of the same clip, plus the oracle's warp of a few frames BY THE GPU'S TRANSFORMS (so that the pixel gate
    tests the warp alone).  Gates: status equal; iteration counts equal; |d(A,B,TX,TY)| <= 1e-4; warped pixels equal."""
    import numpy as np
    from oracle import oracle as O
    status, ts = aw.align()
    P = len(oracle_rec)
    gi = [aw.aligner.info(i) for i in range(P)]
    st_eq = all(bool(status[i]) == bool(oracle_rec[i][0]) for i in range(P))
    reason_eq = all((int(gi[i].fail_reason) == oracle_rec[i][3]) for i in range(P))
    it_eq = all(list(gi[i].iterations[:gi[i].levels]) == oracle_rec[i][2] for i in range(P) if oracle_rec[i][0])
    dmax = 0.0
    for i in range(P):
        g = ts[i]
        dmax = max(dmax, max(abs(a - b) for a, b in zip((g.A, g.B, g.TX, g.TY), oracle_rec[i][1])))
    res = {"frames": P, "status_equal": bool(st_eq and reason_eq), "iterations_equal": bool(it_eq),
           "transform_max_abs_diff": dmax, "transform_tolerance": 1e-4}
    ok = st_eq and reason_eq and it_eq and dmax <= 1e-4
    if aw.warped is not None:
        from oracle import gate as G
        idx = [i for i in warp_frames if i < P]
        O.set_threads(min(16, usable_threads()[0]))
        try:
            want, got = {}, {}
            for name, (gname, oname, _) in WARP_MODES.items():
                aw.warp(ts, getattr(capi, gname))
                torch.cuda.synchronize()
                for i in idx:
                    g = aw.warped[i].cpu().numpy()
                    got[name, i] = g.view(np.uint16) if g.dtype == np.int16 else g
                    want[name, i] = O.bgr_image_warp(frames_host[i], O.Transform.of(ts[i].A, ts[i].B, ts[i].TX, ts[i].TY), getattr(O, oname),
                                                     O.BORDER_CLAMP, max_value=aw.max_value)
            # each GPU mode against its own CPU twin: bit for bit
            for name in WARP_MODES:
                key = "warp_pixels_equal" if name == "exact" else name + "_warp_pixels_equal"
                res[key] = all(bool(np.array_equal(got[name, i], want[name, i])) for i in idx)
                ok = ok and res[key]
            # SURVEY 8(d), integer modes: the GPU's SEPARABLE and CONTRACTED outputs against the UN-contracted oracle (the reference's written order)
            for name in ("separable", "contracted"):
                gate = mode_vs_exact_gate(G, [(got[name, i], want["exact", i]) for i in idx], name)
                gate["frames"] = ["%dx%d frame %d" % (aw.W, aw.H, i) for i in idx]
                res[name + "_vs_exact"] = gate
                ok = ok and gate["pass"]
        finally:
            O.set_threads(1)
        res["warp_frames_checked"] = idx
    res["pass"] = bool(ok)
    res["note"] = ("GPU = the timed path (one %d-frame batch, on-device selection, bgr_image_warp of every frame); CPU = the oracle "
                   "(CPU restatement of the reference, parity unpinned: DESIGN.md section 2) on the first %d frames" % (aw.N, P))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="override the clip length")
    ap.add_argument("--clips-per-gpu", type=int, default=0)
    ap.add_argument("--select", default="device", choices=["host", "device", "stable"],
                    help="device = on-device replica of libstdc++'s nth_element (default, and `value`); stable = VS_SELECT_STABLE, the "
                         "documented STL-independent rule (oracle select rule 1); host = D2H + std::nth_element")
    ap.add_argument("--no-warp", action="store_true", help="alignment only (c2/c3/c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline AND the parity gate that rides on it")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU run); gloo only to rehearse the N>1 flow on one GPU")
    ap.add_argument("--device", type=int, default=-1, help="override LOCAL_RANK -> device (rehearsal on a 1-GPU box)")
    ap.add_argument("--default-levels", action="store_true",
                    help="reference default pyramid_min_width/height = 20 (6 levels at 1080p, 7 at 4K) instead of 256")
    ap.add_argument("--warp-mode", default="separable", choices=["separable", "contracted", "exact", "fast", "sep"],
                    help="separable (= sep) = VS_WARP_LANCZOS2_SEP, the contracted weights and taps summed rows first, then columns (the default "
                         "and `value`: within SURVEY 8(d)'s integer gate of the un-contracted order, re-checked in every run; bit-identical to "
                         "the oracle's separable twin); contracted (= fast) = VS_WARP_LANCZOS2_FAST, round 4's `value`; exact = VS_WARP_LANCZOS2, "
                         "the reference's un-contracted fp32 order")
    ap.add_argument("--phase-correlate", action="store_true", help="aligner with phase_correlate = true (off in the reference's defaults)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (checks the RCCL path)")
    ap.add_argument("--no-roofline-4k", action="store_true", help="skip the isolated 32 x 4K bgr_image_warp measurement")
    ap.add_argument("--exclusive-solver", action="store_true", help="keep the solver kernel in VS_BATCH_EXCLUSIVE mode inside the overlapped step")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the host-resident (PCIe-inclusive) alignment measurement")
    ap.add_argument("--no-c3", action="store_true", help="skip the 4K (configs[2]) leg of the default one-GPU run")
    ap.add_argument("--no-c5", action="store_true", help="skip the 4K 10-bit full-stabilizer (configs[4]) leg of the default one-GPU run")
    ap.add_argument("--c5-clips", type=int, default=8)
    ap.add_argument("--no-drop-in", action="store_true", help="skip the per-frame host-call (drop-in pattern) legs")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 counter passes that measure roofline.traffic in this run (the figure scaled from the "
                         "committed passes is reported instead)")
    ap.add_argument("--c4-strong", action="store_true", help="run the 64-clip strong-scaling leg (configs[3]) on one GPU even if free memory looks short")
    ap.add_argument("--no-c4-strong", action="store_true", help="skip the strong-scaling leg")
    ap.add_argument("--c4-clips", type=int, default=64, help="total clips of the strong-scaling leg (rehearsals use fewer)")
    ap.add_argument("--c4-frames", type=int, default=120)
    args = ap.parse_args()
    default_run = args.workload is None
    if default_run:
        args.workload = "c2"
    args.warp_mode = {"fast": "contracted", "sep": "separable"}.get(args.warp_mode, args.warp_mode)

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
    dist, backend_used, backend_note = None, None, None
    if world > 1 or args.force_dist:
        # one process per GPU; "nccl" is RCCL on ROCm.  No data-path collective: clips are independent.
        # (every rank joins a gloo group first and the ranks AGREE over it whether RCCL is up: all of them report over RCCL, or all
        # of them over gloo with the first failing rank's reason -- video_stabilizer_amd/dist.py)
        # (gloo and RCCL print banners to fd 1 from C++ while they come up: stdout carries ONE JSON line, so fd 1 points at stderr meanwhile)
        sys.stdout.flush()
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            dist, backend_used, backend_note = vsdist.init_with_fallback(args.dist_backend, rank, world,
                                                                          device_id=dev if args.dist_backend == "nccl" else None)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd1, 1)
            os.close(saved_fd1)
        assert dist.get_world_size() == world, "process group has %d ranks, launcher says %d" % (dist.get_world_size(), world)
        if os.environ.get("VS_BENCH_TEST_FAIL_RANK", "") == str(rank):     # test hook (tests/test_bench_gpu.py): this rank dies, the job must end with its code
            sys.stderr.write("bench.py: rank %d fails on purpose (VS_BENCH_TEST_FAIL_RANK)\n" % rank)
            os._exit(7)
        assert world == args.gpus or args.force_dist, "--gpus %d but WORLD_SIZE %d" % (args.gpus, world)
    red_dev = dev if (dist is not None and backend_used == "nccl") else None     # where the three report scalars are reduced
    if dist is not None:
        # roll call BEFORE anything is timed: one line per rank (stderr: stdout carries the one JSON line) with its device, the backend the report
        # collectives run over and the strong-scaling leg's clip split; two ranks on ONE physical device while the node shows a device for each
        # is a launch error (LOCAL_RANK ignored, a bad visible-devices mask) and ends the job with exit code 5 on every rank
        props = torch.cuda.get_device_properties(local_rank)
        # (the key two ranks must not share is host + device INDEX: with every device visible to every rank -- the launch the contract describes -- ranks
        # differ by index; a launcher that shows each rank one device makes them all index 0, and then there are fewer visible devices than ranks and the
        # check below does not apply.  The uuid / PCI id ride along for the reader only: a runtime that reports the same uuid for every device must not
        # be able to end the job)
        me = {"rank": rank, "local_rank": local_rank, "device": local_rank, "device_name": props.name,
              "device_key": "%s/device%d" % (socket.gethostname(), local_rank),
              "device_id": str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or ""),
              "backend": backend_used, "rccl_ranks": dist.get_world_size() if backend_used == "nccl" else 0,
              "c4_clips": vsdist.shard_clips(args.c4_clips, rank, world)}
        everyone = vsdist.roll_call(me)
        sys.stderr.write("bench.py rank %d/%d: device %d (%s, %s), report collectives over %s (%d RCCL ranks), strong-leg clips %s; no 1 -> 8 GPU curve has "
                         "been measured in rounds 1-6\n" % (rank, world, local_rank, props.name, me["device_key"] + " " + me["device_id"], backend_used, me["rccl_ranks"],
                                                            me["c4_clips"] if len(me["c4_clips"]) <= 8 else "%d clips" % len(me["c4_clips"])))
        sys.stderr.flush()
        clash = vsdist.shared_devices(everyone, torch.cuda.device_count()) if args.device < 0 else []
        if clash:
            sys.stderr.write("bench.py rank %d: ranks share a physical device although %d devices are visible: %r\n" % (rank, torch.cuda.device_count(), clash))
            sys.exit(5)

    wl = WORKLOADS[args.workload]
    W, H, bits = wl["w"], wl["h"], wl["bits"]
    n = args.frames or wl["frames"]
    n_clips = args.clips_per_gpu or wl["clips"]
    fmt = capi.FMT_BGR8 if bits == 8 else capi.FMT_BGR10
    params_kw = {} if args.default_levels else dict(pyramid_min_width=256)
    if args.phase_correlate:
        params_kw["phase_correlate"] = 1
    stream = torch.cuda.current_stream()
    # global clip index of local clip j = rank + j*world (clip i -> rank i mod N); textures per rank, path per clip
    seeds = [wl["seed"] + 1000 * (rank + j * world) for j in range(n_clips)]

    def timed_loop(fn, k):
        """k calls of fn between barrier + synchronize on both sides; seconds on this rank, last return value"""
        if dist is not None:
            vsdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            r = fn()
        torch.cuda.synchronize()
        if dist is not None:
            vsdist.barrier()
        return time.perf_counter() - t0, r

    def roofline_of(aw, nframes_total):
        ms = sum(a.elapsed_time(b) for a, b in aw.ev) / len(aw.ev)          # one launch per step over all the rank's frames
        bytes_per_launch = aw.W * aw.H * 3 * 2 * (1 if aw.bits == 8 else 2) * nframes_total  # W*H*3*(in+out) B per frame
        achieved = bytes_per_launch / (ms * 1e-3) / 1e9
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes, scaled to this launch's frame count; null when
        # there is no profile for this frame format
        traffic, tname = None, None
        try:
            tj, tname = load_profile("r05_traffic.json", "r04_traffic.json", "r03_traffic.json")
            key, per = {(1920, 8): ("c2_1080p_240_frames", 240), (3840, 8): ("c3_4k_32_frames", 32)}[(aw.W, aw.bits)]
            traffic = int(tj[key]["traffic_bytes"] / per * nframes_total)
        except Exception:
            pass
        # the instruction count behind "VALU-issue-bound" comes from the committed counter passes of this kernel form, not from a literal
        pmc, pmc_name = load_profile("r05_warp_pmc.json", "r04_warp_pmc.json")
        ipp = pmc.get(args.warp_mode, {}).get("valu_instr_per_px")
        if ipp:
            # at one wave-instruction per 2 cycles per SIMD (the nominal fp32 rate) the frame's instructions alone take t_valu
            t_valu = ipp * aw.W * aw.H * nframes_total / 64.0 / VALU_WAVE_INSTR_PER_S
            bound = "%.1f VALU instructions per output pixel (rocprofv3 SQ_INSTS_VALU, profiles/%s): <= %.2f of the HBM peak at the nominal issue rate of one " \
                    "wave-instruction per 2 cycles per SIMD" % (ipp, pmc_name, bytes_per_launch / t_valu / 1e9 / HBM_PEAK_GBPS)
        else:
            bound = "no committed counter pass for this kernel form"
        return {"kernel": "vs_k_bgr_warp_c3<%s,clamp> (bgr_image_warp)" % WARP_MODES[args.warp_mode][2],
                "bound": "hbm", "binding": "valu",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": "scaled from the committed PMC passes (profiles/%s), not measured in this run" % tname,
                "launch_ms": round(ms, 4), "bytes_per_launch": bytes_per_launch, "valu_instr_per_px": ipp,
                "note": "VALU-issue-bound, not HBM-bound: %s (DESIGN.md section 5); launches overlap the next pass's aligner kernels" % bound}

    if wl["stabilizer"]:
        crop = 32
        factory = synth.TorchClipFactory(W, H, wl["seed"] + 1000 * rank, dev, channels=3, bits=bits)
        all_frames = torch.empty((n_clips * n, H, W, 3), dtype=torch.uint8 if bits == 8 else torch.int16, device=dev)
        clips = [factory.make(n, seeds[j], out=all_frames[j * n:(j + 1) * n])[0] for j in range(n_clips)]
        torch.cuda.synchronize()
        stab = capi.Stabilizer(device=local_rank,                                           # the library default is the reference's bilinear
                               warp_mode=getattr(capi, WARP_MODES[args.warp_mode][0]),
                               select_mode={"device": capi.SELECT_DEVICE, "stable": capi.SELECT_STABLE, "host": capi.SELECT_STL_HOST}[args.select],
                               **params_kw)
        out_buf = torch.empty((n_clips * n, H - 2 * crop, W - 2 * crop, 3), dtype=clips[0].dtype, device=dev)
        aw = None

        def step(timed):
            # all clips of the rank in one call (vs_stabilizer_process_clips): each clip through a fresh stabilizer,
            # alignment and warps of all clips batched together
            r, _ = stab.process_clips_device(all_frames.data_ptr(), n_clips, n, W, H, fmt, out_buf.data_ptr())
            return r
    else:
        aw = AlignWarp(torch, capi, synth, dev, wl, n, n_clips, seeds, params_kw, args, wl["seed"] + 1000 * rank)
        clips = aw.clips

        def step(timed):
            return aw.step(timed)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    # untimed pre-roll (independent of --warmup): steps back to back until >= PREROLL_SECONDS of launches have run, so that the timed loops
    # below see the settled shader clock whatever --warmup was
    preroll_steps, t_pre = 0, time.perf_counter()
    while time.perf_counter() - t_pre < PREROLL_SECONDS:
        for _ in range(4):
            step(False)
        preroll_steps += 4
        torch.cuda.synchronize()
    preroll_s = time.perf_counter() - t_pre
    if aw:
        aw.aligner.enable_timing(True)
    # the driver's loop, REPEATS times back to back; whole-job numbers per repeat: max seconds over ranks, frames summed over ranks (the only
    # collectives of the run)
    reps = []
    for _ in range(REPEATS):
        dt_r, good = timed_loop(lambda: step(True), args.steps)
        reps.append(vsdist.aggregate(dt_r, n * n_clips * args.steps, int(good) * args.steps, device=red_dev))
    shader_mhz = None
    try:
        shader_mhz = round(capi.shader_clock_probe(stream.cuda_stream), 1)      # issued directly behind the timed loops
    except Exception:                                        # noqa: BLE001 -- a diagnostic must not cost the line its headline
        pass
    order = sorted(range(REPEATS), key=lambda i: reps[i][0])
    dt, total_frames, total_good = reps[order[REPEATS // 2]]
    tm = aw.aligner.timings() if aw else None

    align_only = None
    if aw and not args.no_warp:
        # second, separately reported figure: the alignment stages alone (BASELINE configs[1] read literally)
        aw.shared(False)                                       # nothing else runs: one 512-thread workgroup per pair
        aw.aligner.enable_timing(True)
        dt_a, _ = timed_loop(lambda: aw.align(), args.steps)
        dt_a, frames_a, _ = vsdist.aggregate(dt_a, n * n_clips * args.steps, 0, device=red_dev)
        align_only = (dt_a, frames_a, aw.aligner.timings())
        aw.shared(True)

    others = {}
    default_leg = None
    if aw and not args.no_warp:
        # beside `value`: the same step with the OTHER members of the sampler family -- `exact_warp` (VS_WARP_LANCZOS2: the reference's
        # written, un-contracted rounding order; rounds 1-3's `value`), `contracted_warp` (round 4's `value`), `separable_warp`
        for name, (gname, _, _) in WARP_MODES.items():
            if name == args.warp_mode:
                continue
            m = getattr(capi, gname)
            aw.step(False, m)
            dt_f, good_f = timed_loop(lambda: aw.step(False, m), args.steps)
            dt_f, _, good_f = vsdist.aggregate(dt_f, n * n_clips * args.steps, int(good_f) * args.steps, device=red_dev)
            others[name] = (dt_f, good_f)
        # ... and the reference's OWN pipeline on this workload (align + cv::warpAffine's fixed-point bilinear, constant border): `default_warp`
        default_leg = aw.default_warp_leg(timed_loop, args.steps,
                                          aggregate=lambda dt_, fr_, gd_: vsdist.aggregate(dt_, fr_, gd_, device=red_dev))

    stable = None
    if aw and not args.no_warp and args.select == "device":
        # fourth figure: the same step with the selection under the documented STL-independent rule (VS_SELECT_STABLE, SURVEY 8(f)
        # rank 1; bit-identical to the oracle's select rule 1: tests/test_select_stable_gpu.py) -- beside `value`, never as `value`
        aw.aligner.set_select_mode(capi.SELECT_STABLE)
        aw.step(False)
        dt_s, good_s = timed_loop(lambda: aw.step(False), args.steps)
        dt_s, _, good_s = vsdist.aggregate(dt_s, n * n_clips * args.steps, int(good_s) * args.steps, device=red_dev)
        aw.aligner.set_select_mode(capi.SELECT_DEVICE)
        stable = (dt_s, good_s)

    rc = 0
    out = None
    if rank == 0:
        out = {
            "metric": "aligned frames/sec", "value": round((total_good if not wl["stabilizer"] else total_frames) / dt, 2), "unit": "frames/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "ranks": ([{k: e[k] for k in ("rank", "device", "device_key", "device_id", "backend")} for e in everyone] if dist is not None else None),
            "scaling_curve_measured": False,
            "dist_backend": (backend_used if dist is not None else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "value_warp_mode": None if (args.no_warp and not wl["stabilizer"]) else args.warp_mode,
            "value_spread": {"repeats": REPEATS, "value_is": "median",
                             "values": [round((reps[i][2] if not wl["stabilizer"] else reps[i][1]) / reps[i][0], 2) for i in reversed(order)],
                             "ms_per_step": [round(1e3 * reps[i][0] / args.steps, 4) for i in order],
                             "note": "the driver's loop (exactly --steps steps between barrier + synchronize) run %d times back to back; "
                                     "`values` ascending: min, median, max" % REPEATS},
            "preroll": {"steps": preroll_steps, "seconds": round(preroll_s, 4),
                        "note": "untimed steps after --warmup until >= %.2f s of launches have run (clock settling)" % PREROLL_SECONDS},
            "shader_clock_mhz": shader_mhz,
            "vs_baseline": None, "dtype": "u8" if bits == 8 else "u16", "data": "synthetic",
            "config": {"workload": wl["name"] if not args.default_levels else
                       wl["name"].split(",")[0] + ", reference default pyramid (pyramid_min_width = pyramid_min_height = 20)",
                       "frames_per_clip": n, "clips_per_gpu": n_clips, "width": W, "height": H,
                       "bits": bits, "clip_seeds": "clip i -> rank i mod N; path seed %d + 1000 i" % wl["seed"],
                       "selection": "std::nth_element on the host" if args.select == "host"
                       else ("VS_SELECT_STABLE: smallest by (abs_delta, tile index), survivors in tile order (oracle select rule 1)" if args.select == "stable"
                             else "on-device replica of libstdc++ nth_element (same survivors, same order)"),
                       "solver": "clip groups overlapped inside vs_stabilizer_process_clips: the warps of group g run under the alignment of group "
                                 "g + 1 (small-footprint solver build)" if wl["stabilizer"] else
                       "exclusive (512-thread workgroup per pair)" if (args.exclusive_solver or args.no_warp)
                       else "shared (VS_BATCH_SHARED: 256-thread small-footprint build under the previous pass's warp, bit-identical)",
                       "phase_correlate": bool(args.phase_correlate),
                       "warp": None if (args.no_warp and not wl["stabilizer"]) else
                       ("bgr_image_warp lanczos2, un-contracted (VS_WARP_LANCZOS2)" if args.warp_mode == "exact" else
                        "bgr_image_warp %s: <= 1 LSB from the un-contracted order, >= 99.99 %% identical "
                        "(SURVEY 8(d) integer gate, checked in this run: parity.%s_vs_exact)" % (WARP_MODES[args.warp_mode][2], args.warp_mode)),
                       "resident": "HBM",
                       "select_mode_in_force": (aw.aligner.select_mode() if aw else stab.select_mode())},
            "frames_per_step": total_frames // args.steps,
            ("outputs_per_step" if wl["stabilizer"] else "aligned_per_step"): total_good // args.steps,
            "value_counts": ("stabilized output frames + the lag frames that produce none" if wl["stabilizer"] else
                             "aligned frames only (AlignNextFrame true): the first frame of a clip has no predecessor"),
        }
        if backend_note:
            out["dist_backend_note"] = "RCCL could not be initialised (%s): the report scalars were reduced over gloo on the host" % backend_note
        if tm:
            out["stages"] = stage_table(tm, args.steps * REPEATS)
            out["gn_iterations_per_frame"] = round(tm["gn_iterations"] / max(1, tm["frames"]), 2)
        if align_only:
            dt_a, frames_a, tm_a = align_only
            out["align_only"] = {"value": round(frames_a / dt_a, 2), "unit": "frames/s",
                                 "ms_per_step": round(1e3 * dt_a / args.steps, 4), "stages": stage_table(tm_a, args.steps),
                                 "note": "same clips, alignment stages only, all frames counted (no warp launch competing for the CUs; "
                                         "solver kernel in VS_BATCH_EXCLUSIVE mode)"}
        what = {"exact": "VS_WARP_LANCZOS2 = the reference's written fp32 order with no contraction (np.array_equal with the oracle's "
                         "VSO_WARP_LANCZOS2): the figure to compare across rounds (rounds 1-3 reported it as `value`)",
                "contracted": "VS_WARP_LANCZOS2_FAST = the sampler with the multiply-adds fused as the reference's own target allows "
                              "(CMakeLists.txt:151 fma, no strict_float): round 4's `value`",
                "separable": "VS_WARP_LANCZOS2_SEP = the contracted weights and taps summed rows first, then columns"}
        for name, (dt_f, good_f) in others.items():
            out[name + "_warp"] = {"value": round(good_f / dt_f, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt_f / args.steps, 4),
                                   "note": "same step (one loop, after the pre-roll) with bgr_image_warp in " + what[name]}
        if default_leg:
            out["default_warp"] = default_leg
        if stable:
            out["stable_select"] = {"value": round(stable[1] / stable[0], 2), "unit": "frames/s", "ms_per_step": round(1e3 * stable[0] / args.steps, 4),
                                    "note": "same step with VS_SELECT_STABLE: the keep-best-80 % step under a documented STL-independent rule (smallest "
                                            "by (abs_delta, tile index), survivors in tile order) instead of the replica of libstdc++'s nth_element order; "
                                            "bit-identical to the oracle's select rule 1"}
        if aw and aw.ev:
            out["roofline"] = roofline_of(aw, n * n_clips)
            if default_run and world == 1 and not args.no_live_traffic and args.warp_mode != "exact" and not args.no_warp:
                # the HBM-side bytes of the dominant kernel, measured live (child processes; the figure from the committed passes stays
                # beside it, so a regression shows as a disagreement between the two)
                live, how = measure_traffic_live(aw.W, aw.H, n * n_clips, aw.bits, mode=args.warp_mode)
                out["roofline"]["traffic_from_committed_passes"] = out["roofline"]["traffic"]
                if live is not None:
                    out["roofline"]["traffic"] = live
                    out["roofline"]["traffic_source"] = how
                else:
                    out["roofline"]["traffic_source"] += "; a live measurement was attempted and failed: " + how
                if default_leg:
                    # ... and of the reference's own warp (the `default_warp` leg's kernel), same launch shape
                    live_cv, how_cv = measure_traffic_live(aw.W, aw.H, n * n_clips, aw.bits, mode="bilinear_cv")
                    out["default_warp"]["roofline"]["traffic"] = live_cv
                    out["default_warp"]["roofline"]["traffic_source"] = how_cv

    # ---- parity gate + CPU baseline (rank 0 of a one-GPU run; the oracle is the checker, never the thing measured) ----------
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        fh = clips[0][: min(n, 64)].cpu().numpy()
        if bits != 8:
            fh = fh.view("uint16")
        rec = []
        out["cpu_baseline"] = cpu_baseline(fh, params_kw, wl["stabilizer"], record=rec, select_rule=1 if args.select == "stable" else 0)
        out["cpu_baseline"]["cpu_model"] = cpu_model()
        if aw is not None and rec:
            try:
                out["parity"] = parity_gate(torch, capi, aw, params_kw, rec, fh)
                if not out["parity"]["pass"]:
                    rc = 3
            except Exception as e:      # a gate that cannot run is a failed gate
                out["parity"] = {"pass": False, "error": repr(e)}
                rc = 3

    if rank == 0 and not args.no_roofline_4k:
        out["roofline_4k"] = roofline_4k(torch, capi, dev, stream)
    if rank == 0 and not args.no_host_fed and aw is not None:
        out["host_fed"] = host_fed(torch, capi, dev, clips[0], W, H, fmt, params_kw)
    if rank == 0 and world == 1 and not args.no_drop_in and aw is not None:
        try:
            fh = clips[0][: min(n, 48)].cpu().numpy()
            out["drop_in"] = {"%dp" % H: drop_in(capi, dev.index, fh.view("uint16") if bits != 8 else fh, params_kw)}
        except Exception as e:
            out["drop_in"] = {"error": repr(e)}

    # ---- the 4K half of the metric (default one-GPU run): BASELINE configs[2] through the same step --------------------------
    if default_run and world == 1 and not args.no_c3 and aw is not None:
        aw.free()
        aw = None
        clips = None
        try:
            wl3 = WORKLOADS["c3"]
            steps3, n3 = max(2, args.steps), wl3["frames"]
            a3 = AlignWarp(torch, capi, synth, dev, wl3, n3, 1, [wl3["seed"]], params_kw, args, wl3["seed"])
            a3.step(False)
            torch.cuda.synchronize()
            a3.aligner.enable_timing(True)
            dt3, good3 = timed_loop(lambda: a3.step(True), steps3)
            tm3 = a3.aligner.timings()
            others3 = {}
            for name, (gname, _, _) in WARP_MODES.items():
                if name == args.warp_mode:
                    continue
                m3 = getattr(capi, gname)
                a3.step(False, m3)
                dt3f, good3f = timed_loop(lambda: a3.step(False, m3), steps3)
                others3[name + "_warp"] = {"value": round(good3f * steps3 / dt3f, 2), "ms_per_step": round(1e3 * dt3f / steps3, 4)}
            out["c3"] = {"workload": wl3["name"], "value": round(good3 * steps3 / dt3, 2), "unit": "frames/s",
                         "ms_per_step": round(1e3 * dt3 / steps3, 4), "steps": steps3, "frames_per_step": n3, "aligned_per_step": int(good3),
                         "stages": stage_table(tm3, steps3), "gn_iterations_per_frame": round(tm3["gn_iterations"] / max(1, tm3["frames"]), 2),
                         "roofline": roofline_of(a3, n3),
                         "note": "solver in VS_BATCH_SHARED mode; the 4K level 0 (20736 tiles per set) selects on the pair's global scratch"}
            out["c3"].update(others3)
            out["c3"]["default_warp"] = a3.default_warp_leg(timed_loop, steps3)
            if not args.no_cpu_baseline and args.warp_mode != "exact":
                # the 4K frame of the pixel gate: the GPU's warp of frame 1 in the form `value` runs (its measured transform) against the
                # un-contracted oracle
                import numpy as np
                from oracle import oracle as O, gate as G
                st3, ts3 = a3.align()
                a3.warp(ts3, getattr(capi, WARP_MODES[args.warp_mode][0]))
                torch.cuda.synchronize()
                O.set_threads(min(16, usable_threads()[0]))
                try:
                    want3 = O.bgr_image_warp(a3.frames[1].cpu().numpy(), O.Transform.of(ts3[1].A, ts3[1].B, ts3[1].TX, ts3[1].TY), O.WARP_LANCZOS2,
                                             O.BORDER_CLAMP, max_value=255)
                finally:
                    O.set_threads(1)
                g3 = mode_vs_exact_gate(G, [(a3.warped[1].cpu().numpy(), want3)], args.warp_mode)
                gkey = args.warp_mode + "_vs_exact"
                out["c3"][gkey] = {k: g3[k] for k in ("pass", "max_abs_diff_lsb", "least_identical_fraction")}
                if "parity" in out and gkey in out["parity"]:
                    cv = out["parity"][gkey]
                    cv["frames"].append("3840x2160 frame 1")
                    cv["pass"] = bool(cv["pass"] and g3["pass"])
                    cv["max_abs_diff_lsb"] = max(cv["max_abs_diff_lsb"], g3["max_abs_diff_lsb"])
                    cv["least_identical_fraction"] = min(cv["least_identical_fraction"], g3["least_identical_fraction"])
                    out["parity"]["pass"] = bool(out["parity"]["pass"] and g3["pass"])
                if not g3["pass"]:
                    rc = 3
            if not args.no_drop_in and isinstance(out.get("drop_in"), dict) and "error" not in out["drop_in"]:
                out["drop_in"]["2160p"] = drop_in(capi, dev.index, a3.frames[:32].cpu().numpy(), params_kw, calls=32)
            a3.free()
        except Exception as e:
            out["c3"] = {"error": repr(e)}

    # ---- BASELINE configs[4] (default one-GPU run): 4K 10-bit clips through the full stabilizer loop ----------------------------
    if default_run and world == 1 and not args.no_c5:
        if aw is not None:
            aw.free()
            aw = None
            clips = None
        try:
            wl5 = WORKLOADS["c5"]
            W5, H5, n5, nc5, crop5, steps5 = wl5["w"], wl5["h"], wl5["frames"], args.c5_clips, 32, 3
            need5 = 2.2 * nc5 * n5 * W5 * H5 * 3 * 2
            if torch.cuda.mem_get_info(dev)[0] < need5:
                raise RuntimeError("not enough free device memory for %d clips (%.0f GB needed)" % (nc5, need5 / 1e9))
            f5 = synth.TorchClipFactory(W5, H5, wl5["seed"], dev, channels=3, bits=10)
            frames5 = torch.empty((nc5 * n5, H5, W5, 3), dtype=torch.int16, device=dev)
            for j in range(nc5):
                f5.make(n5, wl5["seed"] + 1000 * j, out=frames5[j * n5:(j + 1) * n5])
            del f5
            out5 = torch.empty((nc5 * n5, H5 - 2 * crop5, W5 - 2 * crop5, 3), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()
            res5 = {}
            for key, (gname, _, _) in WARP_MODES.items():
                st5 = capi.Stabilizer(device=local_rank, warp_mode=getattr(capi, gname), **params_kw)

                def step5():
                    return st5.process_clips_device(frames5.data_ptr(), nc5, n5, W5, H5, capi.FMT_BGR10, out5.data_ptr())[0]
                step5()
                dt5, outs5 = timed_loop(step5, steps5)
                res5[key] = (dt5, int(outs5))
                del st5
            dt5, outs5 = res5[args.warp_mode]
            out["c5"] = {"workload": wl5["name"], "clips": nc5, "frames_per_clip": n5, "value": round(nc5 * n5 * steps5 / dt5, 2), "unit": "frames/s",
                         "ms_per_step": round(1e3 * dt5 / steps5, 4), "steps": steps5, "frames_per_step": nc5 * n5, "outputs_per_step": outs5,
                         "dtype": "u16", "warp": "bgr_image_warp %s, crop %d, clamp border" % (WARP_MODES[args.warp_mode][2], crop5),
                         "value_counts": "input frames per second through vs_stabilizer_process_clips (every clip: 10 lag frames without an output)",
                         "note": "BASELINE configs[4] on one GPU: %d of its 64 clips (64 / 8 GPUs); frames and outputs resident in HBM.  Parity at THIS size: one whole "
                                 "60-frame clip against the oracle stabilizer frame for frame with the library's default warp, 14 frames with the separable Lanczos2 "
                                 "(tests/test_configs_gpu.py::test_c5_full_size_clip_matches_the_oracle_stabilizer); the 8-clip batch itself is property-checked "
                                 "(latency pattern, value range, clip independence), the 10-bit pixel gates of the contracted / separable forms are "
                                 "tests/test_warp_gate_gpu.py and tests/test_warp_sep_gpu.py" % nc5}
            for key in WARP_MODES:
                if key != args.warp_mode:
                    out["c5"][key + "_warp"] = {"value": round(nc5 * n5 * steps5 / res5[key][0], 2), "ms_per_step": round(1e3 * res5[key][0] / steps5, 4)}
            # the same loop with the LIBRARY DEFAULTS: cv::warpAffine's fixed-point bilinear (VS_WARP_BILINEAR_CV), black border -- what
            # stabilizer.cpp:97-99 does with every frame (on 10-bit frames: OpenCV's float-weight form, the word-tile kernel)
            st5 = capi.Stabilizer(device=local_rank, **params_kw)

            def step5d():
                return st5.process_clips_device(frames5.data_ptr(), nc5, n5, W5, H5, capi.FMT_BGR10, out5.data_ptr())[0]
            step5d()
            dt5d, _ = timed_loop(step5d, steps5)
            out["c5"]["default_warp"] = {"value": round(nc5 * n5 * steps5 / dt5d, 2), "ms_per_step": round(1e3 * dt5d / steps5, 4),
                                         "warp": "library defaults: VS_WARP_BILINEAR_CV (cv::warpAffine INTER_LINEAR restated), constant border"}
            del st5
            del frames5, out5
            torch.cuda.empty_cache()
        except Exception as e:
            out["c5"] = {"error": repr(e)}
            torch.cuda.empty_cache()

    # ---- BASELINE configs[3] as a strong-scaling leg: 64 clips in total, clip i -> rank i mod N --------------------------------
    # (on one GPU the leg is the N = 1 point of the strong curve: 64 clips = 48 GB of frames + 48 GB of output; it runs when the
    # card has the room, and a failure there must not cost the line its headline)
    want_c4 = default_run and not args.no_c4_strong and not args.no_warp
    if want_c4 and world == 1 and not args.c4_strong:
        need = 2.3 * args.c4_clips * args.c4_frames * 1920 * 1080 * 3
        want_c4 = torch.cuda.mem_get_info(dev)[0] > need
    if want_c4:
        if aw is not None:
            aw.free()
            aw = None
            clips = None
        wl4 = WORKLOADS["c4"]
        mine = vsdist.shard_clips(args.c4_clips, rank, world)
        steps4 = 6          # (the first pass's alignment has no warp to run under: the more steps, the less that start-up weighs)
        try:
            a4 = AlignWarp(torch, capi, synth, dev, wl4, args.c4_frames, len(mine), [wl4["seed"] + i for i in mine], params_kw, args,
                           wl4["seed"] + 1000 * rank)
            a4.step(False)
        except Exception as e:
            if world > 1:
                raise                                    # every rank takes part in the collectives below: fail loudly
            a4 = None
            out["c4_strong"] = {"error": repr(e)}
    if want_c4 and a4 is not None:
        dt4, good4 = timed_loop(lambda: a4.step(False), steps4)
        per_rank = vsdist.gather_seconds(dt4, device=red_dev)
        dt4m, frames4, good4t = vsdist.aggregate(dt4, args.c4_frames * len(mine) * steps4, int(good4) * steps4, device=red_dev)
        if rank == 0:
            out["c4_strong"] = {"workload": wl4["name"], "clips_total": args.c4_clips, "frames_per_clip": args.c4_frames,
                                "clips_per_rank": [len(vsdist.shard_clips(args.c4_clips, r, world)) for r in range(world)],
                                "value": round(good4t / dt4m, 2), "unit": "frames/s", "scaling": "strong", "steps": steps4,
                                "ms_per_step": round(1e3 * dt4m / steps4, 4), "per_rank_seconds": [round(x, 5) for x in per_rank],
                                "note": "total work fixed (BASELINE configs[3]); no data-path collective (grid_search_align.cpp:159-210's "
                                        "independence model); per-rank seconds show a host-side knee if one rank lags"}
        a4.free()

    if rank == 0:
        r4 = out.get("roofline_4k")
        if isinstance(r4, dict) and args.warp_mode in r4 and "roofline" in out:
            # the `value` warp mode where the north star quotes it (4K, isolated), inside the headline's own roofline object
            q = r4[args.warp_mode]
            out["roofline"]["at_4k"] = {"mode": args.warp_mode, "us_per_frame": q["us_per_frame"], "achieved": q["achieved"], "frac": q["frac"],
                                        "frames_per_launch": q["frames_per_launch"]}
        if isinstance(r4, dict):
            # LAST key of the line (a record that keeps only the tail of the line still shows it): {mode: [us per 4K frame, fraction of 8 TB/s]}
            out["roofline_4k_summary"] = dict({k: [v["us_per_frame"], v["frac"]] for k, v in r4.items()},
                                              shader_clock_mhz=shader_mhz, frames_per_launch=32,
                                              note="isolated 32 x 4K launches, HIP events; algorithmic bytes W*H*3*(in+out) per frame; no 1 -> 8 GPU curve has been measured")
        print(json.dumps(out), flush=True)
    if dist is not None:
        vsdist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
