#!/bin/bash
# (A/B of two builds of the library on ONE box: devices differ by several percent, so every comparison alternates the two)
# usage: tools/ab_bench.sh <variant.so> [rounds]   -- alternates the default library and a variant in one run
V="$1"; R="${2:-2}"
run() { python bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = d['align_only']['stages']
print('$1', d['value'], d['ms_per_step'], 'alone: gn', st['gn']['ms_per_step'], 'ingest', st['ingest']['ms_per_step'], 'align_only', d['align_only']['value'])
"; }
for i in $(seq $R); do unset VS_AMD_LIB; run default; export VS_AMD_LIB="$V"; run variant; done
