#!/bin/bash
for r in 1 2 3; do
for e in 0 1; do
  export VS_WARP_HOST_EXTENTS=$e
  a=$(python tools/warp_bench.py --mode lanczos2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  b=$(python tools/warp_bench.py --mode fast 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  c=$(python tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'])")
  echo "round $r host_extents=$e: 4K exact $a | contracted $b | 1080p exact $c"
done; done
