#!/bin/bash
# what a stage of the alignment pass costs beside the warp: builds that launch it twice (-DVS_EXP_REPEAT: 1 ingest, 4 keyframe pass), c2 step
O=gpurun_out/exp19.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --no-c3 --no-c4-strong --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', d['value'], d['ms_per_step'], 'contracted', d['contracted_warp']['ms_per_step'], 'align_only', d['align_only']['ms_per_step'])
" >> $O; }
for r in 1 2 3; do
run base X=1
run ingest_twice VS_AMD_LIB=$V/libvs_amd_rep1.so
run keyframe_twice VS_AMD_LIB=$V/libvs_amd_rep4.so
done
cat $O
