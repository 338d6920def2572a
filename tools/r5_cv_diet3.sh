O=gpurun_out/r5d3; mkdir -p $O
python -m pytest tests/test_warp_cv_gpu.py tests/test_warp_sweep_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do
for v in "" cvold cvfill; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --w 1920 --h 1080 --frames 240 --border constant | tee -a $O/ab.txt
done; done
