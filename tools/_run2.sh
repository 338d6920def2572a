cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
MODES="${MODES:-fast}"
for m in $MODES; do
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/r2/pmcA_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > gpurun_out/r2/pmcA_$m.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/r2/pmcB_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > gpurun_out/r2/pmcB_$m.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/r2/pmcC_$m -- python3 tools/warp_bench.py --mode $m --frames 4 --reps 3 > gpurun_out/r2/pmcC_$m.log 2>&1
done
for m in $MODES; do for p in A B C; do echo "== $m $p"; python3 tools/pmc_summary.py gpurun_out/r2/pmc${p}_$m warp_c3; done; done > gpurun_out/r2/pmc_summary2.txt 2>&1
cat gpurun_out/r2/pmc_summary2.txt
