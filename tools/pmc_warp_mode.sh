#!/bin/bash
# Counter passes (rocprofv3 --pmc, no trace domains) of one warp mode on 4 x 4K frames, and the derived per-pixel / per-wave figures.
# usage (on the GPU box): bash tools/pmc_warp_mode.sh bilinear      -> gpurun_out/pmc_<mode>.json
m=${1:-bilinear}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_$m; rm -rf $O; mkdir -p $O
T="timeout -k 10 300"
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/A -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/A.log 2>&1 &&
$T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/B -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/B.log 2>&1 &&
$T rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC --output-format csv -d $O/C -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/C.log 2>&1
python3 - "$O" "$m" <<'PY'
import collections, csv, glob, json, os, sys
O, m = sys.argv[1], sys.argv[2]
def counters(d):
    agg, ids = collections.defaultdict(float), set()
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bgr_warp_c" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    n = max(1, len(ids))
    return {k: v / n for k, v in agg.items()}
a, b, c = counters("A"), counters("B"), counters("C")
PX4 = 3840 * 2160 * 4
out = {"mode": m, "raw_per_dispatch": {**a, **b, **c}}
if a and b:
    cyc = b["GRBM_GUI_ACTIVE"] / 8
    w = a["SQ_WAVES"]
    out.update({"waves": w, "px_per_wave": PX4 / w, "cycles_per_4_frames": int(cyc),
                "valu_instr_per_wave": round(a["SQ_INSTS_VALU"] / w, 1), "salu_instr_per_wave": round(b["SQ_INSTS_SALU"] / w, 1),
                "lds_instr_per_wave": round(a["SQ_INSTS_LDS"] / w, 1),
                "vmem_rd_per_wave": round(c.get("SQ_INSTS_VMEM_RD", 0) / w, 2), "vmem_wr_per_wave": round(c.get("SQ_INSTS_VMEM_WR", 0) / w, 2),
                "smem_per_wave": round(c.get("SQ_INSTS_SMEM", 0) / w, 2),
                "valu_instr_per_px": round(a["SQ_INSTS_VALU"] * 64 / PX4, 1),
                "simd_cycles_per_wave": round(cyc * 1024 / w, 1),
                "valu_busy_frac": round(a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), 3),
                "wave_cycles_waiting_frac": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
                "wave_cycles_waiting_for_issue_frac": round(a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3),
                "mean_resident_waves_per_simd": round(a["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024), 2),
                "lds_bank_conflict_frac": round(b["SQ_LDS_BANK_CONFLICT"] / max(1.0, b["SQ_LDS_IDX_ACTIVE"]), 3)})
json.dump(out, open("gpurun_out/pmc_%s.json" % m, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "raw_per_dispatch"}))
PY
