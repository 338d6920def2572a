O=gpurun_out/r5d4; mkdir -p $O
python -m pytest tests/test_warp_cv_gpu.py tests/test_warp_sweep_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_bounds.so VS_BOUNDS_BUILD=1 VS_TEST_POISON_ALLOC=165 VS_TEST_HOOKS=1 python -m pytest tests/test_warp_cv_gpu.py tests/test_warp_sweep_gpu.py -x -q -m gpu -k "not 4k_frame" > $O/tests_bounds.log 2>&1 || { tail -40 $O/tests_bounds.log; exit 1; }
tail -1 $O/tests_bounds.log
for rep in 1 2; do
for v in "" cvold; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --bits 16 --frames 16 --border constant | tee -a $O/ab.txt
done; done
