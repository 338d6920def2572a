#!/bin/bash
# VS_WARP_BILINEAR_CV on frames of EQUAL pixel count (8 294 400 = 4K) and different aspect, PURE SUB-PIXEL TRANSLATION (so that the footprint of every tile is
# the same and only the frame's own rim takes the rim path): does the length of the contiguous runs a tile row reads and writes (192 B of a 11 520-B row at
# 4K) change the time per byte?  (zero-code test of "DRAM / L2 locality of 2-D tiles"; the usual rotation about the centre would move the samples of a very
# tall frame by tens of pixels and confound it)
cd "$(dirname "$0")/.."
T=${1:-0.0,0.0,0.3,0.4}
for wh in "128 64800" "256 32400" "512 16200" "960 8640" "1920 4320" "3840 2160" "7680 1080" "15360 540"; do
  set -- $wh
  for bits in 8 16; do
    out=$(timeout -k 10 120 python3 tools/warp_bench.py --mode cv --w $1 --h $2 --frames 32 --reps 20 --bits $bits --transform $T 2>/dev/null | tail -1)
    [ $? -eq 124 ] && { echo "timeout: stopping"; exit 1; }
    echo "w=$1 h=$2 bits=$bits T=$T: $(echo "$out" | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); print(j["us_per_frame_median"], "us", j["GBps_median"], "GB/s", j["frac_of_8TBps"])')"
  done
done
