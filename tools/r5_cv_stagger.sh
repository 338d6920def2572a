O=gpurun_out/r5n; mkdir -p $O
for rep in 1 2; do
for v in "" cvst27 cvst54 cvst108; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}" | tee -a $O/ab.txt
  python tools/warp_bench.py --mode cv --frames 32 --border constant | tee -a $O/ab.txt
done; done
