O=gpurun_out/r5s; mkdir -p $O
for v in "VS_STAB_TIME_CHUNKS=4" "VS_STAB_TIME_CHUNKS=2" "VS_STAB_TIME_CHUNKS=3" "VS_STAB_TIME_CHUNKS=6" "VS_STAB_OVERLAP=0" "VS_STAB_TIME_CHUNKS=4"; do
  echo "== $v" | tee -a $O/chunks.txt
  env $v python tools/stab_long_clip_bench.py 2>&1 | grep "bilinear_cv" | tee -a $O/chunks.txt
done
for v in "VS_STAB_GROUPS=4" "VS_STAB_GROUPS=2" "VS_STAB_GROUPS=8" "VS_STAB_OVERLAP=0" "VS_STAB_CV_SOLVER=1" "VS_STAB_GROUPS=4"; do
  echo "== $v" | tee -a $O/chunks.txt
  env $v python tools/stab_c5_bench.py 2>&1 | grep "frames/s" | tee -a $O/chunks.txt
done
