#!/bin/bash
# the small-footprint solver build at 128 (shipped) / 168 / 228 VGPRs (__launch_bounds__(256, 4 / 3 / 2): 65 / 28 / 0 spilled) inside the c2 step
O=gpurun_out/exp21.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --no-c3 --no-c4-strong --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'], 'gn in step', d['stages']['gn']['ms_per_step'], 'contracted', d['contracted_warp']['ms_per_step'])
" >> $O; }
for r in 1 2 3; do
run vgpr128 X=1
run vgpr168 VS_AMD_LIB=$V/libvs_amd_mwv3.so
run vgpr228 VS_AMD_LIB=$V/libvs_amd_mwv2.so
done
cat $O
