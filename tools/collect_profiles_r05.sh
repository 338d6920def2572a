#!/bin/bash
# Round-5 profile collection on the GPU box (one gpurun call).  Raw output under gpurun_out/r5prof/ (scratch); the summaries are copied into
# profiles/ by `python tools/summarise_profiles.py gpurun_out/r5prof r05` afterwards.
# Counter passes use --pmc alone (no trace domains); the program itself follows `--` (python3, no wrapper).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5prof; rm -rf $O; mkdir -p $O
T="timeout -k 10 600"      # a profiler that aborts can leave its child hanging: bound every pass
# (a) the HEADLINE alone: the c2 step and nothing else, so that the warp kernel's row of the stats file is the timed launches (+ pre-roll) only
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-c3 --no-c5 --no-c4-strong --no-roofline-4k --no-host-fed --no-drop-in --no-cpu-baseline > $O/stats_c2.json 2> $O/stats_c2.err
echo "stats c2 headline rc=$?"
python3 tools/step_trace.py "$(find $O/stats_c2 -name '*kernel_trace.csv' | head -1)" 40 110 > $O/step_trace_shared.md
# (b) the driver's own command, every leg
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $O/stats_default.json 2> $O/stats_default.err
echo "stats default rc=$?"
# (c) counter passes of the warp kernel, the three members of the Lanczos2 family (SQ groups on 4 x 4K frames, FETCH / WRITE on their own on 32 x 4K frames)
for m in lanczos2 fast sep; do
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmcA_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcA_$m.log 2>&1
$T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcB_$m.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcF_$m.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcW_$m.log 2>&1
echo "pmc $m done"
done
# the headline's launch shape (240 x 1080p) in the form `value` runs (separable)
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_c2 -- python3 tools/warp_bench.py --mode sep --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcF_c2.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_c2 -- python3 tools/warp_bench.py --mode sep --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcW_c2.log 2>&1
# (d) the fixed-point bilinear: traffic passes (its SQ passes: tools/pmc_warp_mode.sh cv)
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_cv -- python3 tools/warp_bench.py --mode cv --frames 32 --reps 2 > $O/pmcF_cv.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_cv -- python3 tools/warp_bench.py --mode cv --frames 32 --reps 2 > $O/pmcW_cv.log 2>&1
# (e) BASELINE configs[3] with the host side in C++ beside the Python rank (one GPU: the N = 1 point, two and eight slots on it)
{ apps/bin/vs_many_clips --clips 64 --frames 120 --steps 6; apps/bin/vs_many_clips --clips 64 --frames 120 --steps 6 --devices 0,0;
  apps/bin/vs_many_clips --clips 64 --frames 120 --steps 6 --devices 0,0,0,0,0,0,0,0; apps/bin/vs_many_clips --clips 64 --frames 120 --steps 6 --warp-mode exact; } > $O/many_clips_cpp.jsonl 2> $O/many_clips_cpp.err
{ apps/bin/vs_latency; apps/bin/vs_latency 3840 2160 24; } > $O/latency_cpp.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.db" -size +20M -delete
ls $O
