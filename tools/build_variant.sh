#!/bin/bash
# Builds video_stabilizer_amd/variants/libvs_amd_<name>.so = the library with vs_warp.hip compiled under extra -D flags
# (tuning experiments: select one with VS_AMD_LIB=... ; the other objects come from the regular build).
# usage: tools/build_variant.sh <name> "<flags>" [file.hip ...]   (default file: vs_warp.hip)
#        tools/build_variant.sh bounds      the debug build with bounds-checked LDS / scratch indexing (-DVS_DEBUG_BOUNDS: vs_device.hpp,
#                                           include/vs_amd.h vs_debug_bounds_check) -> variants/libvs_amd_bounds.so, tests/test_bounds_build_gpu.py
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/video_stabilizer_amd/csrc"
if [ "$1" = "bounds" ] && [ $# -eq 1 ]; then set -- bounds "-DVS_DEBUG_BOUNDS" vs_engine.hip vs_warp.hip vs_phase.hip vs_capi.hip; fi
NAME="$1"; FLAGS="$2"; shift 2
FILES="${*:-vs_warp.hip}"
OUTD="$ROOT/video_stabilizer_amd/variants"; BD="$CS/build/variant_$NAME"
mkdir -p "$OUTD" "$BD"
[ -f "$CS/build/vs_engine.o" ] || bash "$CS/build.sh"
objs=""
for o in "$CS"/build/*.o; do
  b="$(basename "$o" .o)"
  skip=0; for f in $FILES; do [ "${f%.*}" = "$b" ] && skip=1; done
  [ $skip = 1 ] || objs="$objs $o"
done
for f in $FILES; do
  extra=""; [ "$f" = "vs_warp.hip" ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $extra $FLAGS -c "$CS/$f" -o "$BD/${f%.*}.o"
  objs="$objs $BD/${f%.*}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUTD/libvs_amd_$NAME.so" $objs
echo "built $OUTD/libvs_amd_$NAME.so"
