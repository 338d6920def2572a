#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value per dispatch."""
import collections
import csv
import glob
import sys

import os
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:   # the newest pass only
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = max(1, len(cnt[k]))
        print(k, "dispatches", n)
        for a, b in sorted(v.items()):
            print("   %-28s %16.0f" % (a, b / n))
