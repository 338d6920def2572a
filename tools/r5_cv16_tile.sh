O=gpurun_out/r5t16; mkdir -p $O
for v in cv16th16 cv16th24; do
  VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so python -m pytest tests/test_warp_cv_gpu.py -x -q -m gpu -k "16 or u16 or bit" > $O/tests_$v.log 2>&1 || { tail -30 $O/tests_$v.log; exit 1; }
  tail -1 $O/tests_$v.log
done
for rep in 1 2; do
for v in "" cv16th16 cv16th24; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --bits 16 --frames 16 --border constant | tee -a $O/ab.txt
done; done
