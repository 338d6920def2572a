mkdir -p gpurun_out/r2
python bench.py --steps 20 --warmup 2 > gpurun_out/r2/bench_c2.json 2> gpurun_out/r2/bench_c2.err; echo "c2 rc=$?"
for wl in c3 c4 c5; do python bench.py --workload $wl --steps 5 --warmup 1 --no-roofline-4k > gpurun_out/r2/bench_$wl.json 2> gpurun_out/r2/bench_$wl.err; echo "$wl rc=$?"; done
python bench.py --default-levels --steps 10 --warmup 1 --no-roofline-4k --no-host-fed > gpurun_out/r2/bench_c2def.json 2>/dev/null
python bench.py --workload c3 --default-levels --steps 5 --warmup 1 --no-roofline-4k --no-host-fed > gpurun_out/r2/bench_c3def.json 2>/dev/null
python - <<'PY'
import json
for wl in ("c2","c3","c4","c5","c2def","c3def"):
    try:
        j=json.loads(open("gpurun_out/r2/bench_%s.json"%wl).read().strip().splitlines()[-1])
        print(wl, "value", j["value"], "ms/step", j["ms_per_step"], "align_only", (j.get("align_only") or {}).get("value"), "fast", (j.get("fast_warp") or {}).get("value"),
              "roof", (j.get("roofline") or {}).get("frac"), "cpu", (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("single_clip",{}).get("value"), "host_fed", (j.get("host_fed") or {}).get("value"), (j.get("host_fed") or {}).get("of_pinned_h2d"), "gn_it", j.get("gn_iterations_per_frame"))
        if wl=="c2": print("  4k:", {k:(v["us_per_frame"], v["frac"]) for k,v in j["roofline_4k"].items()})
    except Exception as e:
        print(wl, "ERR", e)
PY
