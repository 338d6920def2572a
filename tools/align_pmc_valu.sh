#!/bin/bash
# Vector-instruction volume of every kernel of one alignment pass (240 x 1080p frames, device-resident, both solver builds): what the pass costs the
# concurrent warp launch.  usage (GPU box): bash tools/align_pmc_valu.sh  -> gpurun_out/align_valu.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/align_valu; rm -rf $O; mkdir -p $O
for mode in 0 1; do
VS_GN_CORESIDENT=$mode timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/m$mode -- python3 tools/align_pmc.py --frames 240 --reps 2 --device-resident > $O/m$mode.log 2>&1
python3 - "$O/m$mode" "$mode" <<'PY'
import collections, csv, glob, os, sys
d, mode = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
print("VS_GN_CORESIDENT=%s (1 = the small-footprint solver build); per launch, millions of wave-instructions" % mode)
tot = 0
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    m = max(1, len(n[k]))
    print("  %-42s launches %2d  VALU %8.2f M  SALU %7.2f M  LDS %6.2f M  VMEM rd %6.2f M wr %6.2f M  waves %8.0f" % (k, m, c["SQ_INSTS_VALU"] / m / 1e6, c["SQ_INSTS_SALU"] / m / 1e6, c["SQ_INSTS_LDS"] / m / 1e6, c["SQ_INSTS_VMEM_RD"] / m / 1e6, c["SQ_INSTS_VMEM_WR"] / m / 1e6, c["SQ_WAVES"] / m))
PY
done > gpurun_out/align_valu.txt 2>&1
cat gpurun_out/align_valu.txt
