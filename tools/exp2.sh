#!/bin/bash
# warp tile-height variants: isolated 4K timing (exact / contracted), two rounds interleaved on one box
for r in 1 2; do
for v in "" th32 th32rb8 th64 th32rb2; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  e=$(python tools/warp_bench.py --mode lanczos2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  f=$(python tools/warp_bench.py --mode fast 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  h=$(python tools/warp_bench.py --mode lanczos2 --w 1920 --h 1080 --frames 240 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'])")
  echo "round $r variant ${v:-default16}: 4K exact $e | contracted $f | 1080p exact $h"
done; done
unset VS_AMD_LIB
