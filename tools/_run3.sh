mkdir -p gpurun_out/r2
: > gpurun_out/r2/warp_var.txt
for v in "" $VARIANTS; do
  if [ -n "$v" ]; then export VS_AMD_LIB=video_stabilizer_amd/variants/libvs_amd_$v.so; fi
  echo "variant [$v]" >> gpurun_out/r2/warp_var.txt
  python -m pytest tests/test_kernels_gpu.py tests/test_warp_fast_gpu.py -q -m gpu -k "warp" 2>&1 | tail -1 >> gpurun_out/r2/warp_var.txt
  for m in $MODES; do python tools/warp_bench.py --mode $m 2>/dev/null | cut -c1-200 >> gpurun_out/r2/warp_var.txt; done
done
cat gpurun_out/r2/warp_var.txt
