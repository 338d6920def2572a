#!/bin/bash
# round 5: the driver's bench command, then the whole GPU suite with durations (one gpurun call)
set -e
O=gpurun_out/r5b; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { echo "bench rc=$?"; tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5b/bench_default.json"))
for k in ("value","ms_per_step","value_warp_mode","value_spread","preroll","shader_clock_mhz"): print(k, d.get(k))
for k in ("exact_warp","contracted_warp","stable_select"): print(k, d.get(k,{}).get("value"))
print("roofline", {k:d["roofline"][k] for k in ("frac","launch_ms","valu_instr_per_px","traffic")})
print("roofline_4k", {k:(v.get("frac"), v.get("us_per_frame")) for k,v in d["roofline_4k"].items()})
print("parity", {k:(v if not isinstance(v,dict) else (v.get("pass"),v.get("least_identical_fraction"))) for k,v in d["parity"].items() if k!="note"})
for k in ("c3","c5","c4_strong"): print(k, {kk:(vv if not isinstance(vv,dict) else vv.get("value")) for kk,vv in d.get(k,{}).items() if kk in ("value","ms_per_step","exact_warp","contracted_warp","error","separable_vs_exact")})
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
python -m pytest tests -x -q -m gpu --durations=15 > $O/suite.log 2>&1 || { tail -30 $O/suite.log; exit 1; }
tail -25 $O/suite.log
