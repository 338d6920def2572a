#!/bin/bash
# kernel trace of the c2 bench step (steady state) -> gpurun_out/<name>_trace.md ; usage: tools/trace_step.sh <name> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N="$1"; shift
O=gpurun_out/trace_$N; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k "$@" > $O/line.json 2> $O/err.txt
echo "rc=$?"
T=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/step_trace.py $T 60 110 > gpurun_out/${N}_trace.md
S=$(find $O -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/${N}_kernel_stats.csv
cp $O/line.json gpurun_out/${N}_line.json
rm -rf $O
