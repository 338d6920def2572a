#!/bin/bash
# what-if builds of bgr_image_warp (wrong results, timing only): which part of the kernel costs what, at settled clocks
O=gpurun_out/exp17.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  for mode in lanczos2 fast; do
    r=$(env "$@" python3 tools/warp_bench.py --frames 32 --reps 40 --mode $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label $mode: $r" >> $O
  done; }
run base X=1
for v in 32 40 16; do run whatif_$v VS_AMD_LIB=$V/libvs_amd_wi$v.so; done
run base X=1
cat $O
