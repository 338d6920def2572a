# r04: the raw-tile bilinear kernels (u8 and 10-bit) with two tiles per workgroup (variant bil2 = -DVS_WARP_TILES_PER_WG_BILINEAR=2: slower, 24.2 / 20.3 vs
# 22.2 / 16.1 us per 4K frame) and the timing-only what-if builds wi8 (no store) and wi2 (no fill: 8.1 / 6.3 us -- the fill and its memory traffic are
# 14 of the 10-bit kernel's 22 us).  Builds: tools/build_variant.sh bil2 "-DVS_WARP_TILES_PER_WG_BILINEAR=2"; wi8 "-DVS_WARP_WHATIF=8"; wi2 "-DVS_WARP_WHATIF=2".
O=gpurun_out/ab_bil_nt2.log; : > $O
V=video_stabilizer_amd/variants
VS_AMD_LIB=$V/libvs_amd_bil2.so python3 -m pytest tests/test_kernels_gpu.py tests/test_warp_sweep_gpu.py -m gpu -x -q -k "warp" 2>&1 | tail -n 1 >> $O || { cat $O; exit 1; }
run() { local label="$1"; shift
  for args in "--frames 32 --mode bilinear --bits 16" "--frames 32 --mode bilinear"; do
    r=$(env "$@" python3 tools/warp_bench.py --reps 40 $args 2>/dev/null | tail -n 1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['us_per_frame_median'], j['frac_of_8TBps'])")
    echo "$label [$args]: $r" >> $O
  done; }
for r in 1 2; do
  run "base" X=1
  run "2 tiles per workgroup" VS_AMD_LIB=$V/libvs_amd_bil2.so
  run "no store (8)" VS_AMD_LIB=$V/libvs_amd_wi8.so
  run "no fill (2)" VS_AMD_LIB=$V/libvs_amd_wi2.so
done
cat $O
