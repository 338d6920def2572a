#!/bin/bash
# VS_WARP_TILES_PER_WG: a workgroup walks NT consecutive tiles, the next tile's loads in flight during the current tile's sampler blocks
O=gpurun_out/exp20.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  for mode in lanczos2 fast bilinear; do
    r=$(env "$@" python3 tools/warp_bench.py --frames 32 --reps 40 --mode $mode 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label $mode: $r" >> $O
  done; }
for r in 1 2; do
run base X=1
for v in 2 4 8; do run nt_$v VS_AMD_LIB=$V/libvs_amd_nt$v.so; done
done
cat $O
