O=gpurun_out/r5s; mkdir -p $O
for i in 1 2; do for m in 1 0; do
  export VS_STAB_CV_SOLVER=$m
  python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline-4k --no-host-fed 2>$O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 VS_STAB_CV_SOLVER=$m', 'value', d['value'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d.items() if 'warp' in k})
" | tee -a $O/ab.txt
  python tools/stab_long_clip_bench.py 2>&1 | grep "bilinear_cv" | tee -a $O/ab.txt
done; done
