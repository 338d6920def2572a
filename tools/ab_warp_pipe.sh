#!/bin/bash
# r04: the software-pipelined contracted sampler (VS_WARP_FAST_PIPE, vs_warp.hip fast_rows_pipelined) against the row-pair form,
# isolated bgr_image_warp at settled clocks, variants alternating on one box.  Bit-exactness first (the fast-mode tests on the variant).
#   tools/build_variant.sh pipe6 "-DVS_WARP_FAST_PIPE=1 -DVS_WARP_PIPE_AHEAD=6"; ... pipe3 ...; gpurun -- 'bash tools/ab_warp_pipe.sh'
O=gpurun_out/ab_warp_pipe.log; : > $O
V=video_stabilizer_amd/variants
for v in pipe6; do
  VS_AMD_LIB=$V/libvs_amd_$v.so python3 -m pytest tests/test_warp_fast_gpu.py tests/test_warp_gate_gpu.py -m gpu -x -q 2>&1 | tail -2 >> $O || exit 1
done
run() { local label="$1"; shift
  for args in "--frames 32 --mode fast" "--frames 32 --mode fast --bits 16" "--w 1920 --h 1080 --frames 240 --mode fast"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label [$args]: $r us per frame" >> $O
  done; }
for r in 1 2 3; do
  run base X=1
  run pipe6 VS_AMD_LIB=$V/libvs_amd_pipe6.so
  run pipe3 VS_AMD_LIB=$V/libvs_amd_pipe3.so
done
cat $O
