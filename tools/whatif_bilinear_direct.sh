# r04: the tile-less bilinear experiment (VS_WARP_BILINEAR_DIRECT=1) with its loads forced to 4- / 8-byte alignment (wrong results, timing only):
# builds -DVS_WARP_BILINEAR_DIRECT=1 plus -DVS_WARP_WHATIF=64 / 128
O=gpurun_out/whatif_bildirect.log; : > $O
V=video_stabilizer_amd/variants
run() { local label="$1"; shift
  r=$(env "$@" python3 tools/warp_bench.py --reps 40 --frames 32 --mode bilinear 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'])")
  echo "$label bilinear 4K: $r us per frame" >> $O; }
for r in 1 2; do
  run "direct" VS_AMD_LIB=$V/libvs_amd_bildirect.so
  run "direct, 4-aligned loads (64)" VS_AMD_LIB=$V/libvs_amd_bildirect_wi64.so
  run "direct, 8-aligned loads (128)" VS_AMD_LIB=$V/libvs_amd_bildirect_wi128.so
done
cat $O
