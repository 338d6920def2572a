#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== warp occupancy sensitivity (4K x16, us/frame)"
for v in "" pad8k pad21k; do
  if [ -n "$v" ]; then export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_$v.so; else unset VS_AMD_LIB; fi
  echo "variant ${v:-default}: $(python tools/warp_bench.py --mode lanczos2 | tail -1) | fast: $(python tools/warp_bench.py --mode fast | tail -1)"
done
unset VS_AMD_LIB
echo "== align kernel VALU instructions, plain vs cores"
for c in 0 1; do
  export VS_GN_CORESIDENT=$c
  O=gpurun_out/pmc_align_c$c; rm -rf $O
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $O -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O.log 2>&1
  echo "coresident=$c"; python3 tools/pmc_summary.py $O vs_k_align; rm -rf $O
done
unset VS_GN_CORESIDENT
echo "== 96-VGPR co-resident kernel"
export VS_AMD_LIB=$PWD/video_stabilizer_amd/variants/libvs_amd_nt256w5.so
bash tools/ab_cores.sh 1 | grep cores
unset VS_AMD_LIB
bash tools/ab_cores.sh 1
