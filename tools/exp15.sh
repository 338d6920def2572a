#!/bin/bash
# bgr_image_warp 32 x 4K at SETTLED clocks (tools/warp_bench.py now runs >= 80 ms of launches first): the round's tuning variants
# again, alternated on one box.  -> gpurun_out/exp15.log
O=gpurun_out/exp15.log; : > $O
V=video_stabilizer_amd/variants
run() { # label, env...
  local label="$1"; shift
  for mode in lanczos2 fast; do
    r=$(env "$@" python3 tools/warp_bench.py --frames 32 --reps 40 --mode $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'], j['us_per_frame_min'])")
    echo "$label $mode: $r" >> $O
  done
}
for round in 1 2; do
  echo "## round $round" >> $O
  run base X=1
  run host_extents_off VS_WARP_HOST_EXTENTS=0
  for v in th32 mw3 mw5 cf1 fs0 fs2; do run $v VS_AMD_LIB=$V/libvs_amd_$v.so; done
done
cat $O
