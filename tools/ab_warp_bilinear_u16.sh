#!/bin/bash
# r04: bilinear on 16-bit containers with the word tile (VS_WARP_BILINEAR_U16_TILE, the regular build: 64 x 16 output tiles) against the
# float tile (variant u16float = -DVS_WARP_BILINEAR_U16_TILE=0) and against 64 x 32 tiles (u16h32 = -DVS_WARP_TILE_H_BILINEAR_U16=32).
O=gpurun_out/ab_warp_bilinear_u16.log; : > $O
V=video_stabilizer_amd/variants
python3 -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_warp_sweep_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 1 >> $O || { cat $O; exit 1; }
VS_AMD_LIB=$V/libvs_amd_u16h32.so python3 -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_warp_sweep_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 1 >> $O || { cat $O; exit 1; }
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear --bits 16" "--w 1920 --h 1080 --frames 240 --mode bilinear --bits 16"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'], j['frac_of_8TBps'])")
    echo "$label [$args]: $r (us per frame, fraction of 8 TB/s)" >> $O
  done; }
for r in 1 2 3; do
  run "word tile 64x16" X=1
  run "word tile 64x32" VS_AMD_LIB=$V/libvs_amd_u16h32.so
  run "float tile" VS_AMD_LIB=$V/libvs_amd_u16float.so
done
cat $O
