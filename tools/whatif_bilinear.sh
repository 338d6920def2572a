O=gpurun_out/whatif_bilinear.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  r=$(env "$@" python3 tools/warp_bench.py --reps 40 --frames 32 --mode bilinear 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
  echo "$label bilinear 4K: $r us per frame" >> $O; }
for r in 1 2; do
  run base X=1
  run "no fill (2)" VS_AMD_LIB=$V/libvs_amd_wi2.so
  run "no store (8)" VS_AMD_LIB=$V/libvs_amd_wi8.so
  run "loads hit the same lines (32)" VS_AMD_LIB=$V/libvs_amd_wi32.so
  run "no fill, no store (10)" VS_AMD_LIB=$V/libvs_amd_wi10.so
done
cat $O
