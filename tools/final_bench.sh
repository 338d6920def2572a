#!/bin/bash
# every workload of the results table of DESIGN.md section 6 on ONE box, final code -> gpurun_out/bench_lines_final.jsonl
# (copy to profiles/rNN_bench_lines_final.jsonl).  The first line is the driver's own command.
O=gpurun_out/bench_lines_final.jsonl; : > $O
S="--steps 20 --warmup 5"
python bench.py --gpus 1 $S >> $O
Q="--no-host-fed --no-roofline-4k"
python bench.py --workload c3 --steps 10 --warmup 3 $Q >> $O
python bench.py --workload c4 --steps 5 --warmup 2 $Q >> $O
python bench.py --workload c5 --steps 3 --warmup 1 $Q >> $O
python bench.py --workload c2 $S --default-levels $Q --no-cpu-baseline >> $O
python bench.py --workload c3 --steps 10 --warmup 3 --default-levels $Q --no-cpu-baseline >> $O
python bench.py --workload c2 $S --exclusive-solver $Q --no-cpu-baseline >> $O
python bench.py --workload c2 $S --phase-correlate $Q --no-cpu-baseline >> $O
wc -l $O
