"""bench.py contract on the GPU box: one JSON line with the driver's keys, the roofline and cpu_baseline objects;
and the N > 1 flow rehearsed with two gloo ranks sharing the one GPU (the real run uses nccl = RCCL, one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _last_json(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_bench_line_contract(gpu_vs):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--steps", "2", "--warmup", "1",
                          "--c4-clips", "3", "--c4-frames", "6", "--c5-clips", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    j = _last_json(out.stdout)
    assert KEYS <= set(j)
    assert j["metric"] == "aligned frames/sec" and j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 2
    assert j["vs_baseline"] is None and j["dtype"] == "u8" and j["data"] == "synthetic" and j["scaling"] == "weak"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and j["aligned_per_step"] == 7
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["binding"] == "valu" and "traffic_source" in r
    # the HBM-side bytes are measured in the run itself (two rocprofv3 counter passes as child processes) and agree with the
    # algorithmic bytes: no wasted re-reads (8 frames here)
    assert "measured in this run" in r["traffic_source"], r["traffic_source"]
    assert 0.97 * r["bytes_per_launch"] <= r["traffic"] <= 1.10 * r["bytes_per_launch"], (r["traffic"], r["bytes_per_launch"])
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu_model"]
    assert j["align_only"]["value"] > j["value"]
    assert j["rccl_ranks"] == 1
    assert c["all_cores"]["value"] > 0 and c["all_cores"]["cores"] >= c["cores"] and c["usable_threads"] >= 1
    # `value` counts aligned frames only
    assert j["frames_per_step"] == 8 and abs(j["value"] - 7 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-3 * j["value"]
    # the parity gate that rides on the CPU baseline: the timed GPU path against the oracle on the clip's first frames
    p = j["parity"]
    assert p["pass"] is True and p["status_equal"] and p["iterations_equal"] and p["transform_max_abs_diff"] <= 1e-4
    assert p["warp_pixels_equal"] is True and p["contracted_warp_pixels_equal"] is True and p["separable_warp_pixels_equal"] is True and p["frames"] == 8
    # the north star's own operating point: 32 x 4K frames, isolated, every member of the sampler family
    for name in ("exact", "contracted", "separable"):
        q = j["roofline_4k"][name]
        assert q["bound"] == "hbm" and q["unit"] == "GB/s" and q["frames_per_launch"] == 32
        assert abs(q["frac"] - q["achieved"] / q["peak"]) < 1e-3 and q["bytes_per_launch"] == 3840 * 2160 * 3 * 2 * 32
    assert j["roofline_4k"]["separable"]["achieved"] > j["roofline_4k"]["contracted"]["achieved"] > j["roofline_4k"]["exact"]["achieved"]
    # `value` runs the separable sampler and says so; the un-contracted and the contracted figures ride beside it, and the gates that admit
    # the two reassociated forms are in the line
    assert j["value_warp_mode"] == "separable" and "separable" in j["config"]["warp"] and "separable" in r["kernel"]
    assert j["config"]["select_mode_in_force"] == 1
    assert j["exact_warp"]["value"] > 0 and j["contracted_warp"]["value"] > 0 and "separable_warp" not in j and j["stable_select"]["value"] > 0
    # the reference's OWN pipeline on the headline workload (align + cv::warpAffine's fixed-point bilinear, constant border, forward map): alignment-bound,
    # measured with the solver kernel in both batch modes (the faster one leads), with its own in-step roofline and stage table
    dw = j["default_warp"]
    assert dw["value"] > j["value"] and "VS_BORDER_CONSTANT" in dw["warp"] and "gn" in dw["stages"]
    bm = dw["by_solver_mode"]
    assert set(bm) == {"shared", "exclusive"} and dw["value"] == max(bm.values()) == bm[dw["solver"]]
    assert dw["roofline"]["bytes_per_launch"] == 1920 * 1080 * 3 * 2 * 8 and abs(dw["roofline"]["frac"] - dw["roofline"]["achieved"] / 8000.0) < 1e-3
    # ... its HBM-side bytes measured in the run too (the kernel-name filter of the counter passes takes vs_k_bgr_warp_cv_c3: ADVICE r05)
    assert "measured in this run" in dw["roofline"]["traffic_source"], dw["roofline"]["traffic_source"]
    assert 0.95 * dw["roofline"]["bytes_per_launch"] <= dw["roofline"]["traffic"] <= 1.15 * dw["roofline"]["bytes_per_launch"]
    # the record's tail: {mode: [us per 4K frame, fraction of the HBM peak]} as the LAST key of the line, and the `value` mode at 4K inside `roofline`
    assert list(j)[-1] == "roofline_4k_summary" and j["scaling_curve_measured"] is False
    sm = j["roofline_4k_summary"]
    for name in ("separable", "exact", "bilinear_cv", "bilinear_cv_10bit"):
        assert sm[name] == [j["roofline_4k"][name]["us_per_frame"], j["roofline_4k"][name]["frac"]]
    assert r["at_4k"]["mode"] == "separable" and r["at_4k"]["frac"] == sm["separable"][1]
    assert j["roofline_4k"]["bilinear_cv"]["achieved"] > j["roofline_4k"]["bilinear"]["achieved"]
    for key in ("separable_vs_exact", "contracted_vs_exact"):
        g = p[key]
        assert g["pass"] is True and g["max_abs_diff_lsb"] <= 1 and g["least_identical_fraction"] >= 0.9999, key
        assert sum("1920x1080" in f for f in g["frames"]) >= 2
    assert any("3840x2160" in f for f in p["separable_vs_exact"]["frames"])
    # measurement hygiene: an untimed pre-roll of >= 150 ms, the driver's loop three times (value = the median), the shader clock
    sp = j["value_spread"]
    assert sp["repeats"] == 3 and len(sp["values"]) == 3 and sp["values"] == sorted(sp["values"]) and sp["values"][1] == j["value"]
    assert j["preroll"]["seconds"] >= 0.15 and j["preroll"]["steps"] >= 4 and 500 < j["shader_clock_mhz"] < 3500
    # the 4K half of the metric (BASELINE configs[2]) in the same line
    c3 = j["c3"]
    assert "error" not in c3 and c3["value"] > 0 and c3["frames_per_step"] == 120 and c3["aligned_per_step"] == 119
    assert c3["roofline"]["bytes_per_launch"] == 3840 * 2160 * 3 * 2 * 120 and "gn" in c3["stages"]
    assert c3["exact_warp"]["value"] > 0 and c3["contracted_warp"]["value"] > 0 and c3["separable_vs_exact"]["pass"] is True
    assert c3["default_warp"]["value"] > c3["value"] and c3["default_warp"]["roofline"]["bytes_per_launch"] == 3840 * 2160 * 3 * 2 * 120
    # BASELINE configs[4]: 4K 10-bit, full stabilizer loop (2 of the 8 clips per GPU here)
    c5 = j["c5"]
    assert "error" not in c5 and c5["value"] > 0 and c5["dtype"] == "u16" and c5["frames_per_step"] == 2 * 60
    assert c5["outputs_per_step"] == 2 * 50 and c5["exact_warp"]["value"] > 0 and c5["contracted_warp"]["value"] > 0 and "property-checked" in c5["note"] and c5["default_warp"]["value"] > 0
    # the reference's per-frame call pattern on host frames, beside the oracle making the same calls
    for res in ("1080p", "2160p"):
        d = j["drop_in"][res]
        for leg in ("align_next", "process_frame", "process_frame_lanczos2", "cpu_align_next", "cpu_process_frame"):
            assert d[leg]["frames_per_s"] > 0 and d[leg]["ms_per_call"] > 0, (res, leg)
        assert d["align_next"]["results"] == d["align_next"]["calls"] - 1
    assert j["drop_in"]["2160p"]["process_frame"]["results"] == 32 - 10
    # ... and the N = 1 point of the strong-scaling leg (BASELINE configs[3]; 3 clips of 6 frames here)
    s4 = j["c4_strong"]
    assert s4["scaling"] == "strong" and s4["clips_per_rank"] == [3] and s4["value"] > 0 and len(s4["per_rank_seconds"]) == 1
    hf = j["host_fed"]
    assert hf["identical_to_device_resident"] is True and hf["value"] > 0 and 0 < hf["of_pinned_h2d"] <= 1.05


RANKS = 5      # the box's process guard allows six GPU processes at once; this pytest process is one of them


def _spawn(extra_env=None, clips=64, frames=4):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(RANKS), "--frames", "6", "--steps", "2", "--warmup", "1",
                           "--no-cpu-baseline", "--no-roofline-4k", "--no-host-fed", "--dist-backend", "gloo", "--device", "0",
                           "--c4-clips", str(clips), "--c4-frames", str(frames), "--c4-strong"],
                          capture_output=True, text=True, timeout=900, env=env)


def test_bench_spawns_its_own_ranks(gpu_vs):
    """`bench.py --gpus N` with no launcher in the environment starts the ranks itself (gloo here: the ranks share the one GPU of the box;
    the real run is nccl = RCCL with one GPU per rank) and rank 0 reports the whole job.  Five ranks -- as many as the box's process
    guard leaves room for; the eight-rank case of the same launcher loop runs on the CPU (tests/test_dist_cpu.py) -- with BASELINE
    configs[3]'s 64 clips as the strong-scaling leg."""
    out = _spawn()
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                          # one line for the whole job
    assert out.stdout.strip() == lines[0]                           # ... and nothing else on stdout (library banners go to stderr)
    j = json.loads(lines[0])
    assert j["n_gpus"] == RANKS and j["rccl_ranks"] == RANKS and j["dist_backend"] == "gloo"
    # the roll call before anything is timed: one stderr line per rank (device, backend, clip split), the same facts in the line, and the plain
    # statement that no scaling curve has been measured
    for rk in range(RANKS):
        assert "bench.py rank %d/%d: device 0" % (rk, RANKS) in out.stderr, out.stderr[-1500:]
    assert [e["rank"] for e in j["ranks"]] == list(range(RANKS)) and all(e["device"] == 0 and e["backend"] == "gloo" for e in j["ranks"])
    assert j["scaling_curve_measured"] is False and "no 1 -> 8 GPU curve" in out.stderr
    assert j["aligned_per_step"] == RANKS * 5                       # every rank's clip is counted
    assert len(j["value_spread"]["values"]) == 3
    # N > 1, default workload: BASELINE configs[3] as a strong-scaling leg beside the weak `value`: 64 clips, clip i -> rank i mod N,
    # with every rank's seconds
    s = j["c4_strong"]
    assert s["scaling"] == "strong" and s["clips_total"] == 64 and s["clips_per_rank"] == [13, 13, 13, 13, 12] and len(s["per_rank_seconds"]) == RANKS
    assert s["value"] > 0 and all(x > 0 for x in s["per_rank_seconds"])


def test_a_failing_rank_ends_the_job_with_its_exit_code(gpu_vs):
    """rank 3 dies right after the process group is up; the others would wait in the first barrier for the collective's timeout: the
    launcher loop stops them, the job's exit code is the failing rank's, and no report line is printed"""
    import time
    t0 = time.perf_counter()
    out = _spawn({"VS_BENCH_TEST_FAIL_RANK": "3"}, clips=5, frames=4)
    assert out.returncode == 7, (out.returncode, out.stderr[-1500:])
    assert "rank 3 ended with exit code 7" in out.stderr and "{" not in out.stdout
    assert time.perf_counter() - t0 < 300


def test_bench_two_ranks_gloo_rehearsal(gpu_vs):
    port = str(29700 + os.getpid() % 200)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "6", "--steps", "2",
                                       "--warmup", "1", "--no-cpu-baseline", "--no-roofline-4k", "--no-host-fed", "--dist-backend", "gloo", "--device", "0",
                                       "--no-c4-strong"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    j = _last_json(outs[0][0])
    assert j["n_gpus"] == 2 and j["aligned_per_step"] == 2 * 5      # both ranks' clips are counted
    assert "{" not in outs[1][0]                                    # only rank 0 prints
