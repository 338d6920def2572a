#!/bin/bash
# r04: non-temporal output stores in the tuned warp kernels (variant ntstore = -DVS_WARP_NT_STORE=1) against plain stores.
O=gpurun_out/ab_warp_nt_store.log; : > $O
V=video_stabilizer_amd/variants
VS_AMD_LIB=$V/libvs_amd_ntstore.so python3 -m pytest tests/test_kernels_gpu.py tests/test_warp_sweep_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 1 >> $O || { cat $O; exit 1; }
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear --bits 16" "--frames 32 --mode bilinear" "--frames 32 --mode fast" "--w 1920 --h 1080 --frames 240 --mode fast"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'], j['frac_of_8TBps'])")
    echo "$label [$args]: $r" >> $O
  done; }
for r in 1 2 3; do
  run "plain" X=1
  run "non-temporal" VS_AMD_LIB=$V/libvs_amd_ntstore.so
done
cat $O
