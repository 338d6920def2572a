O=gpurun_out/r5j; mkdir -p $O
VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_cvm8.so python -m pytest tests/test_warp_cv_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do
  unset VS_AMD_LIB
  echo "variant default" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
for v in cvm8 cvm4; do
for nt in 1 2 4 8; do
  export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so
  echo "variant $v nt=$nt" | tee -a $O/ab.txt
  VS_WARP_CV_TILES_PER_WG=$nt python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
done; done; done
