#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/r2prof/, scratch) into the committed summaries
under profiles/: kernel-stats CSVs of the bench runs, and rNN_warp_pmc.json / rNN_traffic.json, which bench.py reads for the
counter-derived fields of its roofline objects.  usage: python tools/summarise_profiles.py [gpurun_out/r3prof] [r03]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r2prof")
DST = os.path.join(ROOT, "profiles")
RN = sys.argv[2] if len(sys.argv) > 2 else "r02"        # round prefix of the files written
SECOND = "fast" if RN == "r02" else "contracted"        # round 3: VS_WARP_LANCZOS2_FAST is the contracted form of the sampler
VALUE_FORM = "separable" if RN >= "r05" else (SECOND if RN >= "r04" else "exact")   # round 4: bench.py's `value` runs the contracted form, round 5: the separable one
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def counters(d, kernel="warp_c3"):
    agg, ids = collections.defaultdict(float), set()
    # (gpurun merges every call's output into the local scratch directory: only the newest pass of a directory counts)
    files = sorted(glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
    n = max(1, len(ids))
    return {k: v / n for k, v in agg.items()}


for wl in ("c2", "c3", "c5", "default"):
    found = sorted(glob.glob(os.path.join(SRC, "stats_" + wl, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not found:
        continue
    f = found[-1]
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "vs_k_" in r[0] or "vsp" in r[0]]     # the library's kernels (torch's generator kernels dropped)
    with open(os.path.join(DST, RN + "_bench_%s_kernel_stats.csv" % wl), "w", newline="") as o:
        csv.writer(o, quoting=csv.QUOTE_ALL).writerows(keep)
    shutil.copy(os.path.join(SRC, "stats_%s.json" % wl), os.path.join(DST, RN + "_bench_%s_line.json" % wl))

PX4, PX32 = 3840 * 2160 * 4, 3840 * 2160 * 32
out = {"_comment": "vs_k_bgr_warp_c3<u8> on MI355X, rocprofv3 --pmc passes of tools/warp_bench.py (tools/collect_profiles.sh): SQ / GRBM "
                   "passes on 4 x 4K frames per dispatch, FETCH_SIZE and WRITE_SIZE in passes of their own on 32 x 4K frames.  "
                   "FETCH_SIZE is doubled (gfx950 counts 128-byte read requests at 64 bytes: MI355X_MICROARCH.md, re-checked on "
                   "this kernel's 12-byte-per-lane loads by tools/calibrate_counters.py in round 1: 0.500x); WRITE_SIZE is exact.  "
                   "valu_frac = SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the counter charges one "
                   "quad-cycle per instruction, so a stream of 2-cycle scalar fp32 instructions reads above 1 (the fast kernel); "
                   "cycles_per_valu_instr = SIMD cycles of the dispatch / VALU instructions, to hold against the issue costs of "
                   "profiles/r02_ubench_mix.txt.  "
                   "lds_frac = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)."}
for name, m in (("exact", "lanczos2"), (SECOND, "fast")) + ((("separable", "sep"),) if RN >= "r05" else ()):
    a, b = counters("pmcA_" + m), counters("pmcB_" + m)
    fz, wz = counters("pmcF_" + m)["FETCH_SIZE"], counters("pmcW_" + m)["WRITE_SIZE"]
    cyc = b["GRBM_GUI_ACTIVE"] / N_XCD
    out[name] = {
        "valu_instr_per_px": round(a["SQ_INSTS_VALU"] * 64 / PX4, 1),
        "lds_instr_per_px": round(a["SQ_INSTS_LDS"] * 64 / PX4, 1),
        "salu_instr_per_wave": round(b["SQ_INSTS_SALU"] / a["SQ_WAVES"], 1),
        "cycles_per_4_frames": int(cyc),
        "valu_frac": round(a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * N_SIMD), 3),
        "cycles_per_valu_instr": round(cyc * N_SIMD / a["SQ_INSTS_VALU"], 2),
        "lds_frac": round(b["SQ_LDS_IDX_ACTIVE"] / (cyc * N_CU), 3),
        "lds_bank_conflict_frac": round(b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"], 3),
        "wave_cycles_waiting_frac": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
        "fetch_size_kib_raw_32_frames": int(fz), "write_size_kib_32_frames": int(wz),
        "traffic_bytes_per_frame": int((2 * fz + wz) * 1024 / 32),
        "algorithmic_bytes_per_frame": 3840 * 2160 * 3 * 2,
    }
json.dump(out, open(os.path.join(DST, RN + "_warp_pmc.json"), "w"), indent=1)

fz, wz = counters("pmcF_c2")["FETCH_SIZE"], counters("pmcW_c2")["WRITE_SIZE"]
tr = {"_comment": "HBM-side traffic of vs_k_bgr_warp_c3<u8,lanczos2" + (" separable" if RN >= "r05" else (" contracted" if RN >= "r04" else "")) + ",clamp> per launch of 240 x 1080p frames (" + RN + " kernel), separate "
                  "--pmc FETCH_SIZE / WRITE_SIZE passes, counter unit KiB, FETCH_SIZE doubled (see " + RN + "_warp_pmc.json).",
      "c2_1080p_240_frames": {"fetch_size_kib_raw": int(fz), "write_size_kib": int(wz), "traffic_bytes": int((2 * fz + wz) * 1024),
                              "algorithmic_bytes": 1920 * 1080 * 3 * 2 * 240},
      "c3_4k_32_frames": {"fetch_size_kib_raw": out[VALUE_FORM]["fetch_size_kib_raw_32_frames"], "write_size_kib": out[VALUE_FORM]["write_size_kib_32_frames"],
                          "traffic_bytes": out[VALUE_FORM]["traffic_bytes_per_frame"] * 32, "algorithmic_bytes": 3840 * 2160 * 3 * 2 * 32,
                          "kernel_form": VALUE_FORM}}
try:
    fz, wz = counters("pmcF_cv", "bgr_warp_c")["FETCH_SIZE"], counters("pmcW_cv", "bgr_warp_c")["WRITE_SIZE"]
    tr["bilinear_cv_4k_32_frames"] = {"fetch_size_kib_raw": int(fz), "write_size_kib": int(wz), "traffic_bytes": int((2 * fz + wz) * 1024),
                                      "algorithmic_bytes": 3840 * 2160 * 3 * 2 * 32, "kernel": "vs_k_bgr_warp_cv_c3 (VS_WARP_BILINEAR_CV)"}
except Exception:
    pass
json.dump(tr, open(os.path.join(DST, RN + "_traffic.json"), "w"), indent=1)
for f in ("host_fed_1080p.json", "host_fed_4k.json", "latency_1080p.json", "latency_4k.json", "latency_cpp.txt", "step_trace_shared.md",
          "step_trace_exclusive.md", "pmc_align_summary.txt", "stats_c2x.json", "many_clips_cpp.jsonl"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, RN + "_" + f))
for f, t in (("ubench_valu.txt", "r02_ubench_valu.txt"), ("ubench_mix.txt", "r02_ubench_mix.txt")):
    p = os.path.join(ROOT, "gpurun_out", "r2", f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, t))
print(json.dumps(out, indent=1))
print(json.dumps(tr, indent=1))
