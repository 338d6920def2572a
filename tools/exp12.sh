#!/bin/bash
# Is the warp kernel clock-limited?  Samples rocm-smi's shader clock / power while bgr_image_warp runs back to back for a few seconds,
# per mode; and an idle sample before.  -> gpurun_out/exp12.log
O=gpurun_out/exp12.log; : > $O
echo "## idle" >> $O
rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|fclk|mclk|power" >> $O
for mode in lanczos2 fast bilinear; do
  echo "## mode $mode" >> $O
  python3 tools/warp_bench.py --frames 32 --reps 400 --mode $mode > gpurun_out/exp12_$mode.json 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|power \(W\)|Socket Power|Graphics Package" | tr '\n' ' ' >> $O; echo >> $O; sleep 0.3; done
  wait $pid
  cat gpurun_out/exp12_$mode.json | tail -1 >> $O
done
cat $O
