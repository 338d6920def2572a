#!/bin/bash
# A/B on ONE box: the c2 bench step with the small-footprint aligner kernel (default for full batches) and without it
R="${1:-2}"
run() { python bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --steps 20 --warmup 5 --no-c3 $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d['align_only']['stages']
print('$1', d['value'], 'ms/step', d['ms_per_step'], 'in-step gn', d['stages']['gn']['ms_per_step'], 'warp launch', d['roofline']['launch_ms'], '| alone: gn', st['gn']['ms_per_step'], 'align_only', d['align_only']['value'], 'contracted', d['contracted_warp']['value'])
"; }
for i in $(seq $R); do run plain "--exclusive-solver $2"; run cores "$2"; done
