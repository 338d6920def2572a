#!/bin/bash
# VS_SELECT_STABLE against VS_SELECT_DEVICE: one AlignNextFrame at a time (C++), and the c2 / c3 steps
O=gpurun_out/exp22.log; : > $O
for r in 1 2; do
  for m in 1 2; do apps/bin/vs_latency 1920 1080 48 256 24 $m 2>/dev/null | tail -1 >> $O; done
  for m in 1 2; do apps/bin/vs_latency 3840 2160 24 256 24 $m 2>/dev/null | tail -1 >> $O; done
done
run() { python3 bench.py --no-roofline-4k --no-host-fed --no-c3 --no-c4-strong --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['value'], d['ms_per_step'], 'gn in step', d['stages']['gn']['ms_per_step'], 'alone', d['align_only']['stages']['gn']['ms_per_step'], 'align_only', d['align_only']['value'], 'parity', d.get('parity', {}).get('pass'), 'iters', d['gn_iterations_per_frame'])
" >> $O; }
for r in 1 2; do run --select device; run --select stable; done
run --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --select device
run --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --select stable
cat $O
