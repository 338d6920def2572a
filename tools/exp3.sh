#!/bin/bash
# (historical: VS_INGEST_WGS existed in the build this experiment ran on; the ingest kernel has since moved to a 2-D grid)
# bounded ingest grid under the shared-mode c2 step: ms/step, warp launch, ingest stage time; and ingest alone
run() { python bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --steps 20 --warmup 5 --no-c3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d['align_only']['stages']
print('VS_INGEST_WGS=$1', d['value'], 'ms/step', d['ms_per_step'], 'in-step ingest', d['stages']['ingest']['ms_per_step'], 'gn', d['stages']['gn']['ms_per_step'], 'warp launch', d['roofline']['launch_ms'], '| alone: ingest', st['ingest']['ms_per_step'], 'contracted', d['contracted_warp']['value'])
"; }
for r in 1 2; do for g in 0 256 512 1024 2048 4096; do export VS_INGEST_WGS=$g; run $g; done; done
