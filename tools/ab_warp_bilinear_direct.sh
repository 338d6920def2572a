#!/bin/bash
# builds: tools/build_variant.sh bildirect "-DVS_WARP_BILINEAR_DIRECT=1"; tools/build_variant.sh bildirect_wi8 "-DVS_WARP_BILINEAR_DIRECT=1 -DVS_WARP_WHATIF=8"
# r04: the tile-less bilinear path (VS_WARP_BILINEAR_DIRECT: interior tiles fetch each pixel's 2 x 2 window with two unaligned 8-byte
# loads through the vector L1) against the byte-tile path (the regular build; variant "bildirect" = -DVS_WARP_BILINEAR_DIRECT=1).  Bit-exactness first.
O=gpurun_out/ab_warp_bilinear_direct.log; : > $O
V=video_stabilizer_amd/variants
VS_AMD_LIB=$V/libvs_amd_bildirect.so python3 -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 3 >> $O || { cat $O; exit 1; }
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear" "--w 1920 --h 1080 --frames 240 --mode bilinear"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label [$args]: $r us per frame" >> $O
  done; }
for r in 1 2 3; do
  run tile X=1
  run direct VS_AMD_LIB=$V/libvs_amd_bildirect.so
  run "direct, no store (8)" VS_AMD_LIB=$V/libvs_amd_bildirect_wi8.so
done
cat $O
