for p in 1 0 1 0; do
VS_ALIGNER_STREAM_PRIORITY=$p python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-host-fed --no-roofline-4k 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prio $p', j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['stages']['gn'], j['fast_warp']['value'])
"
done
