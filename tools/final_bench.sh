#!/bin/bash
# all bench lines of the results table in one box
O=gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/c2.json 2> $O/c2.err; echo c2 $?
python bench.py --workload c3 --steps 10 --warmup 2 --no-host-fed > $O/c3.json 2> $O/c3.err; echo c3 $?
python bench.py --workload c4 --steps 10 --warmup 2 --no-host-fed --no-roofline-4k > $O/c4.json 2> $O/c4.err; echo c4 $?
python bench.py --workload c5 --steps 5 --warmup 1 --no-host-fed --no-roofline-4k > $O/c5.json 2> $O/c5.err; echo c5 $?
python bench.py --default-levels --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k > $O/c2_default.json 2> $O/c2_default.err; echo c2d $?
python bench.py --workload c3 --default-levels --steps 10 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k > $O/c3_default.json 2> $O/c3_default.err; echo c3d $?
