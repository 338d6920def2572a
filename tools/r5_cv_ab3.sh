O=gpurun_out/r5k; mkdir -p $O
python -m pytest tests/test_warp_cv_gpu.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2 3; do
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --w 1920 --h 1080 --frames 240 --border constant | tee -a $O/ab.txt
done
python tools/warp_bench.py --mode cv --frames 32 --border clamp --transform 0.01,-0.02,3.3,-2.7 | tee -a $O/ab.txt
