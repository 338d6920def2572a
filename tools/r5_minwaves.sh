O=gpurun_out/r5m; mkdir -p $O
run() { python bench.py --no-cpu-baseline --no-roofline-4k --no-host-fed --no-c4-strong --no-c3 --no-c5 --no-drop-in --no-live-traffic --steps 20 --warmup 5 2>$O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'value', d['value'], 'ms', d['ms_per_step'], 'gn under warp', d['stages']['gn']['ms_per_step'], 'cv', d['bilinear_cv_warp']['by_solver_mode'], 'exact', d['exact_warp']['value'])
" | tee -a $O/ab.txt; }
for i in 1 2; do
  unset VS_AMD_LIB; run default
  export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_minw3.so; run minw3
done
