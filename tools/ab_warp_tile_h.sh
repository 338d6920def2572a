#!/bin/bash
# r04: output tile height of the tuned warp kernels (VS_WARP_TILE_H: 16 regular, variants bilh32 / bilh64 = -DVS_WARP_TILE_H=32 / 64):
# the per-workgroup prologue (tile geometry, fill set-up) is paid once per tile, i.e. per 4 / 8 / 16 pixels of a thread.
O=gpurun_out/ab_warp_tile_h.log; : > $O
V=video_stabilizer_amd/variants
for v in bilh32 bilh64; do
  VS_AMD_LIB=$V/libvs_amd_$v.so python3 -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 1 >> $O || { cat $O; exit 1; }
done
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear" "--w 1920 --h 1080 --frames 240 --mode bilinear" "--frames 32 --mode bilinear --bits 16" "--frames 32 --mode fast" "--frames 32 --mode lanczos2"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label [$args]: $r us per frame" >> $O
  done; }
for r in 1 2; do
  run h16 X=1
  run h32 VS_AMD_LIB=$V/libvs_amd_bilh32.so
  run h64 VS_AMD_LIB=$V/libvs_amd_bilh64.so
done
cat $O
