#!/bin/bash
run() { python bench.py --workload c3 --no-cpu-baseline --no-roofline-4k --no-host-fed --steps 10 --warmup 3 $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c3 $1', d['value'], 'ms/step', d['ms_per_step'], 'in-step gn', d['stages']['gn']['ms_per_step'], 'warp launch', d['roofline']['launch_ms'], '| align_only', d['align_only']['value'], 'gn alone', d['align_only']['stages']['gn']['ms_per_step'], 'contracted', d['contracted_warp']['value'])
"; }
for i in 1 2 3; do run exclusive --exclusive-solver; run shared ""; done
