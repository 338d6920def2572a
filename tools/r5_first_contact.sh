set -e
mkdir -p gpurun_out/r5a
python -m pytest tests/test_warp_sep_gpu.py tests/test_warp_fast_gpu.py -x -q -m gpu -s > gpurun_out/r5a/tests.log 2>&1 || { tail -40 gpurun_out/r5a/tests.log; exit 1; }
tail -5 gpurun_out/r5a/tests.log
./tools/bin/check_rcp | tee gpurun_out/r5a/check_rcp.txt
for rep in 1 2; do
for m in fast sep lanczos2; do python tools/warp_bench.py --mode $m --frames 32 | tee -a gpurun_out/r5a/warp.jsonl; done
done
python tools/warp_bench.py --mode fast --bits 16 --frames 16 | tee -a gpurun_out/r5a/warp.jsonl
python tools/warp_bench.py --mode sep --bits 16 --frames 16 | tee -a gpurun_out/r5a/warp.jsonl
python tools/warp_bench.py --mode fast --w 1920 --h 1080 --frames 120 | tee -a gpurun_out/r5a/warp.jsonl
python tools/warp_bench.py --mode sep --w 1920 --h 1080 --frames 120 | tee -a gpurun_out/r5a/warp.jsonl
