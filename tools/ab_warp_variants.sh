#!/bin/bash
# A/B of warp-kernel builds on ONE box (devices differ by several percent): the regular library and any number of variants of
# tools/build_variant.sh, alternating, two passes.  The tests of the mode run against every variant first.
# usage (on the GPU box): bash tools/ab_warp_variants.sh "<warp_bench.py arguments>" "<pytest selection>" variant1 [variant2 ...]
#   e.g.  bash tools/ab_warp_variants.sh "--mode cv --frames 32 --border constant" "tests/test_warp_cv_gpu.py" cvold
A="$1"; T="$2"; shift 2
O=gpurun_out/ab_warp; mkdir -p $O
for v in "" "$@"; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  python -m pytest $T -x -q -m gpu > $O/tests_${v:-default}.log 2>&1 || { tail -30 $O/tests_${v:-default}.log; exit 1; }
  echo "${v:-default}: $(tail -1 $O/tests_${v:-default}.log)"
done
for rep in 1 2; do for v in "" "$@"; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py $A | tee -a $O/ab.txt
done; done
