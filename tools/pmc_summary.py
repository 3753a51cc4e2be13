#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel-name substring."""
import csv, collections, glob, sys
pat, sub = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(pat)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k:30s} n={len(v):3d} mean={sum(v) / len(v):.5g}")
