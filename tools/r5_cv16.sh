O=gpurun_out/r5l; mkdir -p $O
python -m pytest tests/test_warp_cv_gpu.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do
  python tools/warp_bench.py --mode cv --bits 16 --frames 16 --border constant | tee -a $O/ab.txt
  python tools/warp_bench.py --mode bilinear --bits 16 --frames 16 --border constant | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
done
