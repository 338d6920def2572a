#!/bin/bash
# Builds video_stabilizer_amd/libvs_amd.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: fp32/fp64 round exactly as written (DESIGN.md "Numerics"); FMAs are explicit.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libvs_amd.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 ${VS_EXTRA_FLAGS:-} -O3 ${VS_RESOURCE_REPORT:+-Rpass-analysis=kernel-resource-usage} -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
SRCS="vs_kernels.hip vs_warp.hip vs_phase.hip vs_capi.hip vs_engine.hip vs_host.cpp"
mkdir -p "$HERE/build"
objs=""
pids=""
for f in $SRCS; do
  [ -f "$HERE/$f" ] || continue
  o="$HERE/build/${f%.*}.o"
  objs="$objs $o"
  if [ ! -f "$o" ] || [ "$HERE/$f" -nt "$o" ] || [ -n "$(find "$HERE" -maxdepth 1 \( -name '*.hpp' -o -name '*.inc' \) -newer "$o" -print -quit)" ] \
     || [ "$HERE/../../include/vs_amd.h" -nt "$o" ]; then
    # vs_warp.hip: the SLP vectorizer would re-pack its scalar fp32 chains into v_pk_* (slower on gfx950, see the file header)
    extra=""; [ "$f" = "vs_warp.hip" ] && extra="-fno-slp-vectorize"
    case "$f" in
      *.hip) "$HIPCC" $FLAGS $extra -c "$HERE/$f" -o "$o" & pids="$pids $!" ;;
      *.cpp) "$HIPCC" -x hip $FLAGS -c "$HERE/$f" -o "$o" & pids="$pids $!" ;;
    esac
  fi
done
for p in $pids; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" $objs
echo "built $OUT"
# link-level drop-in for the reference's sixteen Halide AOT symbols (include/vs_halide_abi.h): plain host C++ on libvs_amd.so
g++ -std=c++17 -O2 -fPIC -Wall -shared -o "$HERE/../libvs_halide_abi.so" "$HERE/vs_halide_abi.cpp" -L"$HERE/.." -lvs_amd '-Wl,-rpath,$ORIGIN'
echo "built $HERE/../libvs_halide_abi.so"
