#!/usr/bin/env python3
"""Table of the library's kernels in a rocprofv3 --kernel-trace of bench.py (profiles/r02_step_trace.md):
start / end in microseconds from the first one.  usage: python tools/step_trace.py <kernel_trace.csv> [first_row last_row]"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "vs_k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(rows))
print("| kernel | queue | start us | end us | duration us |\n|---|---|---|---|---|")
for r in rows[lo:hi]:
    name = re.search(r"vs_k_\w+(<[^(]*>)?", r["Kernel_Name"]).group(0)
    name = re.sub(r"<unsigned char[^>]*>", "<u8>", name)
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("| %s | %s | %.0f | %.0f | %.0f |" % (name, r["Queue_Id"], s, e, e - s))
