# r04: what do rim tiles cost?  Variant wi256 (-DVS_WARP_WHATIF=256) fills EVERY tile the way rim tiles are filled (per-item border tests and
# clamps, right results).  Answer: nothing measurable (4K contracted 37.6-37.8 vs 37.8-38.3 us; 1080p 9.50 vs 9.50-9.56; bilinear 3.9 vs 3.8).
O=gpurun_out/whatif_rim.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  for args in "--frames 32 --mode fast" "--w 1920 --h 1080 --frames 240 --mode fast" "--w 1920 --h 1080 --frames 240 --mode lanczos2" "--w 1920 --h 1080 --frames 240 --mode bilinear"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 30 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
    echo "$label [$args]: $r us per frame" >> $O
  done; }
for r in 1 2; do
  run base X=1
  run "all tiles rim (256)" VS_AMD_LIB=$V/libvs_amd_wi256.so
done
cat $O
