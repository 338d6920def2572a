#!/bin/bash
# Round-3 profile collection on the GPU box (one gpurun call).  Raw output under gpurun_out/r3prof/ (scratch); the summaries
# are copied into profiles/ by `python tools/summarise_profiles.py gpurun_out/r3prof r03` afterwards.
# Counter passes use --pmc alone (no trace domains); the program itself follows `--` (python3, no wrapper).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3prof; rm -rf $O; mkdir -p $O
T="timeout -k 10 500"      # a profiler that aborts can leave its child hanging: bound every pass
# the driver's own command (c2 step, c3 leg, parity gate, CPU baseline, roofline_4k, host_fed) under the kernel trace
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/stats_c2.json 2> $O/stats_c2.err
echo "stats default rc=$?"
python3 tools/step_trace.py "$(find $O/stats_c2 -name '*kernel_trace.csv' | head -1)" 40 110 > $O/step_trace_shared.md
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-roofline-4k > $O/stats_c3.json 2> $O/stats_c3.err
echo "stats c3 rc=$?"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2x -- python3 bench.py --steps 20 --warmup 5 --exclusive-solver --no-c3 --no-cpu-baseline --no-host-fed --no-roofline-4k > $O/stats_c2x.json 2> $O/stats_c2x.err
python3 tools/step_trace.py "$(find $O/stats_c2x -name '*kernel_trace.csv' | head -1)" 40 80 > $O/step_trace_exclusive.md
for m in lanczos2 fast; do
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmcA_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcA_$m.log 2>&1
$T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcB_$m.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcF_$m.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcW_$m.log 2>&1
echo "pmc $m done"
done
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_c2 -- python3 tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcF_c2.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_c2 -- python3 tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcW_c2.log 2>&1
# the solver kernel of the c2 batch, both builds (VS_GN_CORESIDENT forces one): instructions, waiting, memory instructions
for c in 0 1; do
  export VS_GN_CORESIDENT=$c
  $T rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $O/pmc_align_c$c -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmc_align_c$c.log 2>&1
  $T rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmcB_align_c$c -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmcB_align_c$c.log 2>&1
  { echo "== VS_GN_CORESIDENT=$c"; python3 tools/pmc_summary.py $O/pmc_align_c$c vs_k_align; python3 tools/pmc_summary.py $O/pmcB_align_c$c vs_k_align; } >> $O/pmc_align_summary.txt 2>&1
done
unset VS_GN_CORESIDENT
python3 tools/host_fed_bench.py > $O/host_fed_1080p.json 2>/dev/null
python3 tools/host_fed_bench.py 4k > $O/host_fed_4k.json 2>/dev/null
python3 tools/latency_stages.py > $O/latency_1080p.json 2>/dev/null
python3 tools/latency_stages.py 4k > $O/latency_4k.json 2>/dev/null
{ apps/bin/vs_latency; apps/bin/vs_latency 3840 2160 24; apps/bin/vs_latency 1920 1080 48 256 24 2; apps/bin/vs_latency 3840 2160 24 256 24 2; } > $O/latency_cpp.txt 2>&1
# keep the merged scratch small: the raw traces are tens of MB
find $O -name "*kernel_trace.csv" -size +20M -delete
ls $O
