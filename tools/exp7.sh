#!/bin/bash
# how long does the strong-scaling leg take at the clip counts a rank sees at N = 8, 4, 2 (8, 16, 32 clips per rank)?
for c in 8 16 32; do
  t0=$(date +%s.%N)
  python bench.py --c4-strong --c4-clips $c --no-cpu-baseline --no-roofline-4k --no-host-fed --no-c3 --steps 20 --warmup 5 2> gpurun_out/exp7_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['c4_strong'])
"
  t1=$(date +%s.%N)
  echo "c4_strong with $c clips on one rank: $(echo "$t1 - $t0" | bc) s wall"
done
