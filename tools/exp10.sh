#!/bin/bash
for r in 1 2 3; do
for v in "" cf1; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  a=$(python tools/warp_bench.py --mode lanczos2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  b=$(python tools/warp_bench.py --mode fast 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_frame_median'], d['us_per_frame_min'])")
  echo "round $r variant ${v:-default}: 4K exact $a | contracted $b"
done; done
