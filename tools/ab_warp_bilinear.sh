#!/bin/bash
# r04: the bilinear bgr_image_warp (the stabilizer's default mode, imgproc.cpp:472 INTER_LINEAR) with workgroups that walk 2 / 4 tiles,
# the next tile's loads in flight under the current tile's sampler blocks (VS_WARP_TILES_PER_WG_BILINEAR).  Bit-exactness first.
O=gpurun_out/ab_warp_bilinear.log; : > $O
V=video_stabilizer_amd/variants
for v in bil2 bil4; do
  VS_AMD_LIB=$V/libvs_amd_$v.so python3 -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -1 >> $O || exit 1
done
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear" "--frames 32 --mode bilinear --bits 16" "--w 1920 --h 1080 --frames 240 --mode bilinear"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label [$args]: $r us per frame" >> $O
  done; }
for r in 1 2 3; do
  run base X=1
  run bil2 VS_AMD_LIB=$V/libvs_amd_bil2.so
  run bil4 VS_AMD_LIB=$V/libvs_amd_bil4.so
done
cat $O
