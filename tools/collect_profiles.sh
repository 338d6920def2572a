#!/bin/bash
# Round-2 profile collection on the GPU box (one gpurun call).  Raw output under gpurun_out/r2prof/, summaries are copied
# into profiles/ by tools/summarise_profiles.py afterwards.  Counter passes use --pmc alone (no trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; rm -rf $O; mkdir -p $O
T="timeout -k 10 400"      # a profiler that aborts can leave its child hanging: bound every pass
B="--steps 20 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 bench.py $B > $O/stats_c2.json 2> $O/stats_c2.err
echo "stats c2 rc=$?"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k > $O/stats_c3.json 2> $O/stats_c3.err
echo "stats c3 rc=$?"
for m in lanczos2 fast; do
$T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmcA_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcA_$m.log 2>&1
$T rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmcB_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > $O/pmcB_$m.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcF_$m.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_$m -- python3 tools/warp_bench.py --mode $m --frames 32 --reps 2 > $O/pmcW_$m.log 2>&1
echo "pmc $m done"
done
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_c2 -- python3 tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcF_c2.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW_c2 -- python3 tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 --reps 2 > $O/pmcW_c2.log 2>&1
# the alignment stages of the c2 batch (device-resident frames: one launch per stage)
$T rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $O/pmcA_align -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmcA_align.log 2>&1
$T rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmcB_align -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmcB_align.log 2>&1
$T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmcC_align -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmcC_align.log 2>&1
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF_align -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/pmcF_align.log 2>&1
for p in A B C F; do echo "== align $p"; python3 tools/pmc_summary.py $O/pmc${p}_align vs_k_; done > $O/pmc_align_summary.txt 2>&1
echo "pmc align done"
python3 tools/calibrate_counters.py > $O/calib.txt 2>&1 || true
python3 tools/host_fed_bench.py > $O/host_fed_1080p.json 2>/dev/null
python3 tools/host_fed_bench.py 4k > $O/host_fed_4k.json 2>/dev/null
python3 tools/latency_stages.py > $O/latency_1080p.json 2>/dev/null
python3 tools/latency_stages.py 4k > $O/latency_4k.json 2>/dev/null
{ apps/bin/vs_latency; apps/bin/vs_latency 3840 2160 24; VS_GN_HELPERS=1 VS_GN_PIPELINE=0 VS_GN_POLL=0 apps/bin/vs_latency; } > $O/latency_cpp.txt 2>&1
for p in A B F W; do for m in lanczos2 fast; do echo "== $m $p"; python3 tools/pmc_summary.py $O/pmc${p}_$m warp_c3; done; done > $O/pmc_summary.txt 2>&1
for p in F W; do echo "== c2 $p"; python3 tools/pmc_summary.py $O/pmc${p}_c2 warp_c3; done >> $O/pmc_summary.txt 2>&1
find $O -name "*kernel_stats.csv" | head
ls $O
