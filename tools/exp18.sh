#!/bin/bash
# L2 warm-up of a later tile's source lines (VS_WARP_PREFETCH_DIST): distance sweep at settled clocks
O=gpurun_out/exp18.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  for mode in lanczos2 fast; do
    r=$(env "$@" python3 tools/warp_bench.py --frames 32 --reps 40 --mode $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label $mode: $r" >> $O
  done; }
for r in 1 2; do
run base X=1
for v in 64 128 256; do run pf_$v VS_AMD_LIB=$V/libvs_amd_pf$v.so; done
done
cat $O
