O=gpurun_out/stab_groups.log; : > $O
for r in 1 2; do
for c in 8 4 2; do
  v=$(VS_STAB_GROUPS=$c python3 bench.py --workload c5 --steps 4 --warmup 1 --no-cpu-baseline --no-host-fed --no-roofline-4k --no-drop-in 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "VS_STAB_GROUPS=$c c5: $v" >> $O
done
done
cat $O
