#!/bin/bash
# A/B on one box: the library as built vs a variant, c2 with the driver's flags (no c3 / c4 legs), alternated
V="$1"; R="${2:-3}"; O=gpurun_out/exp16.log; : > $O
run() { python3 bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --no-c3 --no-c4-strong --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d['align_only']['stages']; si = d['stages']
print('$1', d['value'], d['ms_per_step'], 'in step: ingest', si['ingest']['ms_per_step'], 'gn', si['gn']['ms_per_step'], '| alone: ingest', st['ingest']['ms_per_step'], 'gn', st['gn']['ms_per_step'], 'align_only', d['align_only']['value'], 'contracted', d['contracted_warp']['value'])
" >> $O; }
for i in $(seq $R); do unset VS_AMD_LIB; run default; export VS_AMD_LIB="$V"; run variant; done
cat $O
