set -e
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests/test_warp_cv_gpu.py tests/test_warp_sep_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for rep in 1 2; do
for m in cv bilinear sep; do python tools/warp_bench.py --mode $m --frames 32 | tee -a $O/warp.jsonl; done
done
python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/warp.jsonl
python tools/warp_bench.py --mode cv --w 1920 --h 1080 --frames 120 | tee -a $O/warp.jsonl
python tools/warp_bench.py --mode cv --bits 16 --frames 16 | tee -a $O/warp.jsonl
