#!/bin/bash
# The fixed-point bilinear at the roofline_4k launch shape, two clocks side by side on ONE box: the HIP-event figure of tools/warp_bench.py (the whole call:
# table kernel + warp kernel + the gaps between stream operations) and rocprofv3's AverageNs of the warp kernel alone, for the regular library and variants.
# usage: tools/cv_kernel_vs_events.sh [variant ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cv_kve; rm -rf $O; mkdir -p $O
for pass in 1 2; do for v in regular "$@"; do for bits in 8 16; do
  lib=$PWD/video_stabilizer_amd/libvs_amd.so; [ $v = regular ] || lib=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so
  ev=$(VS_AMD_LIB=$lib timeout -k 10 120 python3 tools/warp_bench.py --mode cv --frames 32 --reps 20 --bits $bits 2>/dev/null | grep '^{"kernel"' | tail -1)
  VS_AMD_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${v}_${bits}_$pass -- python3 tools/warp_bench.py --mode cv --frames 32 --reps 20 --bits $bits > $O/${v}_${bits}_$pass.log 2>&1
  f="$(find $O/${v}_${bits}_$pass -name '*kernel_stats.csv' | head -1)"
  python3 - "$f" "$ev" "$v" "$bits" "$pass" <<'PY'
import csv, json, sys
f, ev, v, bits, p = sys.argv[1:6]
e = json.loads(ev)
nbytes = 3840 * 2160 * 3 * 2 * (int(bits) // 8) * 32
k = [r for r in csv.DictReader(open(f)) if "bgr_warp_cv_c3" in r["Name"]][0]
t = [r for r in csv.DictReader(open(f)) if "cv_tables" in r["Name"]]
avg = float(k["AverageNs"])
print("pass %s %-8s %2s-bit: events %.2f us/frame = %.4f | kernel alone (rocprofv3 AverageNs / 32) %.2f us/frame = %.4f, min %.2f%s"
      % (p, v, bits, e["us_per_frame_median"], e["frac_of_8TBps"], avg / 32e3, nbytes / avg / 8000.0, float(k["MinNs"]) / 32e3,
         (" | table kernel %.1f us per launch" % (float(t[0]["AverageNs"]) / 1e3)) if t else ""))
PY
done; done; done
