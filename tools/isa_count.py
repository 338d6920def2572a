#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing: per basic block (label to label) the number of VALU / SALU /
LDS / VMEM instructions and the most frequent opcodes.  usage: python tools/isa_count.py file.s <kernel-substring> [-v]
(make the listing with: hipcc --offload-arch=gfx950 -O3 <flags of csrc/build.sh> -S --cuda-device-only -o file.s file.hip)"""
import collections
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
verbose = "-v" in sys.argv
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
        continue
    op = t.split()[0]
    blocks[cur].append(op)
    if op.startswith(("s_cbranch", "s_branch")):         # a fall-through path behind a branch is a block of its own
        cur = cur.split("+")[0] + "+%d" % (len(blocks) + 1)
        blocks[cur] = []


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


tot = collections.Counter()
for name, ops in blocks.items():
    c = collections.Counter(kind(o) for o in ops)
    tot.update(c)
    if len(ops) >= (1 if verbose else 30):
        pk = sum(1 for o in ops if o.startswith("v_pk_"))
        top = collections.Counter(o for o in ops if o.startswith("v_")).most_common(14)
        print("%-14s %5d instr: valu %4d (packed %3d) salu %4d lds %3d vmem %3d | %s" % (name, len(ops), c["valu"], pk, c["salu"], c["lds"], c["vmem"],
              " ".join("%s:%d" % (o.replace("v_", ""), n) for o, n in top)))
print("total", dict(tot))
