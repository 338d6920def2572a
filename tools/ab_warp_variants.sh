#!/bin/bash
# THE A/B driver for warp-kernel builds, on ONE box (devices differ by several percent): the regular library and any number of variants of
# tools/build_variant.sh (video_stabilizer_amd/variants/libvs_amd_<name>.so -- built HERE before the call, or on the box by the caller), alternating.
# The tests of the mode run against every variant first (skip with "" as the test selection).
# usage (on the GPU box): bash tools/ab_warp_variants.sh "<warp_bench.py arguments>" "<pytest selection>" variant1 [variant2 ...]
#   e.g.  bash tools/ab_warp_variants.sh "--mode cv --frames 32 --border constant" "tests/test_warp_cv_gpu.py" r05
# PASSES=3 for three alternating passes; BITS="8 16" runs every pass at both depths.
# (The one-off scripts of rounds 4-5 -- ab_warp_bilinear / _u16 / _tile_h / _nt_store / _pipe, ab_bilinear_nt2, whatif_bilinear / _rim, ab_bench, ab_cores,
# ab_stab_groups / _time_chunks, ab_stream_priority -- were calls of this driver or of an environment knob with fixed arguments; their answers are in
# profiles/r04_ab_*.md, r04_warp_pipeline.md, r05_warp_cv.md, r05_stab_cv_solver.txt, and the scripts are in history.)
A="$1"; T="$2"; shift 2
O=gpurun_out/ab_warp; mkdir -p $O
if [ -n "$T" ]; then
  for v in "" "$@"; do
    if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
    timeout -k 10 600 python -m pytest $T -x -q -m gpu > $O/tests_${v:-default}.log 2>&1 || { tail -30 $O/tests_${v:-default}.log; exit 1; }
    echo "${v:-default}: $(tail -1 $O/tests_${v:-default}.log)"
  done
fi
for rep in $(seq 1 ${PASSES:-2}); do for v in "" "$@"; do for bits in ${BITS:-default}; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  B=""; [ "$bits" = default ] || B="--bits $bits"
  out=$(timeout -k 10 180 python tools/warp_bench.py $A $B | tail -1); rc=$?
  [ $rc -eq 124 ] && { echo "timeout in variant ${v:-default}: stopping"; exit 1; }
  echo "pass $rep ${v:-default} $B: $out" | tee -a $O/ab.txt
done; done; done
