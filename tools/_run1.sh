mkdir -p gpurun_out/r2
python -m pytest tests/test_kernels_gpu.py tests/test_warp_fast_gpu.py -q -m gpu > gpurun_out/r2/t1.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/t1.txt
tail -4 gpurun_out/r2/t1.txt
for m in lanczos2 fast bilinear; do python tools/warp_bench.py --mode $m; done > gpurun_out/r2/warp_v3.txt 2>&1
python tools/warp_bench.py --mode lanczos2 --bits 16 >> gpurun_out/r2/warp_v3.txt 2>&1
python tools/warp_bench.py --mode fast --bits 16 >> gpurun_out/r2/warp_v3.txt 2>&1
python tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 64 >> gpurun_out/r2/warp_v3.txt 2>&1
python tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 1 >> gpurun_out/r2/warp_v3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r2/warp_v3.txt | cut -c1-250
