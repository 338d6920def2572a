#!/bin/bash
# r04: does the priority of the aligner's stream change the overlapped c2 / c3 step?  (VS_ALIGNER_STREAM_PRIORITY=low|high, read once)
O=gpurun_out/ab_stream_priority.log; : > $O
F="--steps 20 --warmup 5 --no-c5 --no-c4-strong --no-roofline-4k --no-host-fed --no-drop-in --no-cpu-baseline --no-live-traffic"
for r in 1 2 3; do
  for p in default low high; do
    v=$( ( [ $p = default ] && python3 bench.py $F || VS_ALIGNER_STREAM_PRIORITY=$p python3 bench.py $F ) 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['c3']['value'], j['c3']['ms_per_step'])")
    echo "$p: c2 value, ms/step, warp launch ms | c3 value, ms/step: $v" >> $O
  done
done
cat $O
