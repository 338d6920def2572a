# r04: is anything idle in c5?  Kernel trace of bench.py --workload c5, union of the kernel intervals over the timed half: 99.2 % busy (0.56 ms of gaps in 66 ms).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c5trace; rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --no-host-fed --no-roofline-4k --no-drop-in > $O/line.json 2> $O/err.txt
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/c5trace/t/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "vs_k" in r["Kernel_Name"] or "vsp" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows)
# the last third of the trace = timed steps
t0 = ev[len(ev) // 2][0]
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
import collections
tot = collections.Counter()
for s, e, n in ev: tot[n.split("(")[0][-40:]] += e - s
print("span ms", span / 1e6, "busy frac", busy / span, "n kernels", len(ev))
print("largest gaps (us):", sorted((g[0] / 1e3 for g in gaps), reverse=True)[:12])
print("sum of gaps ms", sum(g[0] for g in gaps) / 1e6)
for k, v in tot.most_common(6): print(k, round(v / 1e6, 2), "ms")
PY
