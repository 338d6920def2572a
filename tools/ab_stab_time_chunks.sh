O=gpurun_out/stab_time_chunks.log; : > $O
for r in 1 2; do
for c in 8 6 4 3 2; do
  echo "VS_STAB_TIME_CHUNKS=$c" >> $O
  VS_STAB_TIME_CHUNKS=$c python3 tools/stab_long_clip_bench.py 2>/dev/null | grep contracted >> $O
done
done
cat $O
