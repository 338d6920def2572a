#!/bin/bash
# Round-6 profile collection on the GPU box (one gpurun call).  Raw output under gpurun_out/r6prof/ (scratch); the summaries this script
# writes into gpurun_out/r6prof/keep/ are what gets copied into profiles/ (names r06_*).
# Counter passes use --pmc alone (no trace domains); the program itself follows `--` (python3, no wrapper).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6prof; rm -rf $O; mkdir -p $O/keep
T="timeout -k 10 600"      # a profiler that aborts can leave its child hanging: bound every pass
# (a) the HEADLINE alone: the c2 step and nothing else, so that the warp kernel's row of the stats file is the timed launches (+ pre-roll) only
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-c3 --no-c5 --no-c4-strong --no-roofline-4k --no-host-fed --no-drop-in --no-cpu-baseline > $O/keep/r06_bench_c2_line.json 2> $O/stats_c2.err
echo "stats c2 headline rc=$?"
cp "$(find $O/stats_c2 -name '*kernel_stats.csv' | head -1)" $O/keep/r06_bench_c2_kernel_stats.csv
python3 tools/step_trace.py "$(find $O/stats_c2 -name '*kernel_trace.csv' | head -1)" 40 110 > $O/keep/r06_step_trace_shared.md
# (b) the driver's own command, every leg
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $O/keep/r06_bench_default_line.json 2> $O/stats_default.err
echo "stats default rc=$?"
cp "$(find $O/stats_default -name '*kernel_stats.csv' | head -1)" $O/keep/r06_bench_default_kernel_stats.csv
# (c) ONE kernel-stats file per warp mode and depth for the roofline_4k launch shape (32 x 4K frames per launch): bytes / AverageNs can be recomputed per kernel
for spec in "sep 8" "fast 8" "lanczos2 8" "cv 8" "cv 16" "bilinear 8" "bilinear 16"; do
  set -- $spec
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4k_$1_$2 -- python3 tools/warp_bench.py --mode $1 --frames 32 --reps 20 --bits $2 > $O/r4k_$1_$2.log 2>&1
  f="$(find $O/r4k_$1_$2 -name '*kernel_stats.csv' | head -1)"
  [ -n "$f" ] && cp "$f" $O/keep/r06_roofline4k_$1_$2bit_kernel_stats.csv
  grep '^{"kernel"' $O/r4k_$1_$2.log | tail -1 > $O/keep/r06_roofline4k_$1_$2bit_events.json
  echo "roofline_4k stats $1 $2-bit done"
done
# (d) traffic + instruction counters of the fixed-point bilinear (8- and 10-bit), FETCH / WRITE on their own
for bits in 8 16; do
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_cv$bits -- python3 tools/warp_bench.py --mode cv --frames 32 --reps 2 --bits $bits > $O/pmcF_cv$bits.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_cv$bits -- python3 tools/warp_bench.py --mode cv --frames 32 --reps 2 --bits $bits > $O/pmcW_cv$bits.log 2>&1
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmcA_cv$bits -- python3 tools/warp_bench.py --mode cv --frames 4 --reps 3 --bits $bits > $O/pmcA_cv$bits.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, json, os, sys
O = sys.argv[1]
W, H, FR = 3840, 2160, 32
rows = []
for f in sorted(glob.glob(os.path.join(O, "keep", "r06_roofline4k_*_kernel_stats.csv"))):
    tag = os.path.basename(f)[len("r06_roofline4k_"):-len("_kernel_stats.csv")]
    bits = 16 if tag.endswith("16bit") else 8
    nbytes = W * H * 3 * 2 * (bits // 8) * FR
    for r in csv.DictReader(open(f)):
        if "bgr_warp_c" in r["Name"]:
            avg = float(r["AverageNs"])
            rows.append((tag, r["Name"][:70], int(r["Calls"]), avg, nbytes / avg, nbytes / avg / 8000.0))
ev = {}
for f in glob.glob(os.path.join(O, "keep", "r06_roofline4k_*_events.json")):
    try:
        ev[os.path.basename(f)[len("r06_roofline4k_"):-len("_events.json")]] = json.loads(open(f).read())
    except Exception:
        pass
def pmc(d, ctr, frames):
    vals, ids = 0.0, set()
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "bgr_warp_c" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
    return vals / len(ids) if ids else None
with open(os.path.join(O, "keep", "r06_roofline4k.md"), "w") as out:
    out.write("# roofline_4k launch shape (32 x 3840x2160 per launch): rocprofv3 --kernel-trace --stats per mode, beside the HIP-event figure of the same process\n\n")
    out.write("| mode | kernel | calls | AverageNs | bytes / AverageNs (GB/s) | of 8 TB/s | HIP events in the same run: us per frame, fraction |\n|---|---|---|---|---|---|---|\n")
    for tag, name, calls, avg, gbps, frac in rows:
        e = ev.get(tag, {})
        out.write("| %s | `%s` | %d | %.0f | %.1f | %.4f | %s, %s |\n" % (tag, name, calls, avg, gbps, frac, e.get("us_per_frame_median"), e.get("frac_of_8TBps")))
    out.write("\nbytes = W*H*3*(in+out) per frame x 32 (SURVEY 8d); the stats rows pool the warm-up launches of tools/warp_bench.py with the timed ones (same shape).\n")
    out.write("The fixed-point bilinear's launch is preceded by vs_k_cv_tables (its own row in the stats files: ~4 us per 32-frame launch); the HIP-event figure brackets the\n"
              "whole call -- table kernel, the dependent-launch gap, warp kernel -- and is therefore ~3 % above the warp kernel's own AverageNs (tools/cv_kernel_vs_events.sh).\n\n")
    for bits in (8, 16):
        fz, wz = pmc("pmcF_cv%d" % bits, "FETCH_SIZE", 32), pmc("pmcW_cv%d" % bits, "WRITE_SIZE", 32)
        iv, wv = pmc("pmcA_cv%d" % bits, "SQ_INSTS_VALU", 4), pmc("pmcA_cv%d" % bits, "SQ_WAVES", 4)
        if fz and wz:
            alg = W * H * 3 * 2 * (bits // 8) * 32
            out.write("fixed-point bilinear %d-bit: FETCH_SIZE %.0f KiB (x2: gfx950 tallies 128-byte requests at 64) + WRITE_SIZE %.0f KiB = %.0f bytes per launch = %.3f x algorithmic\n"
                      % (bits, fz, wz, (2 * fz + wz) * 1024, (2 * fz + wz) * 1024 / alg))
        if iv:
            out.write("fixed-point bilinear %d-bit: SQ_INSTS_VALU %.0f per launch of 4 frames = %.1f vector instructions per 64 output pixels\n" % (bits, iv, iv / (W * H * 4 / 64.0)))
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.db" -size +20M -delete
ls $O/keep
