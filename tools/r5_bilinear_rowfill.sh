O=gpurun_out/r5rf; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_warp_sweep_gpu.py -x -q -m gpu -k "warp" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do
for v in "" rowfill0; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode bilinear --frames 32 | tee -a $O/ab.txt
  python tools/warp_bench.py --mode bilinear --bits 16 --frames 16 | tee -a $O/ab.txt
done; done
