#!/bin/bash
run() { python bench.py --workload c5 --no-cpu-baseline --no-roofline-4k --no-host-fed --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 VS_STAB_OVERLAP=$1', d['value'], 'ms/step', d['ms_per_step'], 'outputs', d['outputs_per_step'])
"; }
for i in 1 2; do export VS_STAB_OVERLAP=0; run 0; export VS_STAB_OVERLAP=1; run 1; done
