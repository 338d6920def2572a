O=gpurun_out/r5c3; mkdir -p $O
run() { python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-4k --no-host-fed $2 2>$O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'value', d['value'], 'ms', d['ms_per_step'], 'gn', d['stages']['gn']['ms_per_step'], 'cv', d['bilinear_cv_warp'].get('by_solver_mode'), 'exact', d['exact_warp']['value'])
" | tee -a $O/ab.txt; }
for i in 1 2; do run shared ""; run exclusive "--exclusive-solver"; done
