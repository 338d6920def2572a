#!/bin/bash
# A/B of VS_WARP_BILINEAR_CV builds on ONE box: alternating passes of tools/warp_bench.py (32 x 4K per launch, HIP events, settled clocks) over the
# regular library and the variants named on the command line (video_stabilizer_amd/variants/libvs_amd_<name>.so), 8-bit and 10-bit.
# usage: tools/ab_cv.sh [passes] name...        e.g.  tools/ab_cv.sh 3 r05
set -u
cd "$(dirname "$0")/.."
passes=${1:-2}; shift || true
for p in $(seq 1 "$passes"); do
  for name in regular "$@"; do
    lib=$PWD/video_stabilizer_amd/libvs_amd.so
    [ "$name" = regular ] || lib=$PWD/video_stabilizer_amd/variants/libvs_amd_$name.so
    for bits in 8 16; do
      out=$(VS_AMD_LIB=$lib timeout -k 10 120 python3 tools/warp_bench.py --mode cv --frames 32 --reps 20 --bits $bits 2>/dev/null | tail -1)
      rc=$?
      [ $rc -eq 124 ] && { echo "timeout in $name: stopping"; exit 1; }
      echo "pass $p $name bits=$bits: $out"
    done
  done
done
