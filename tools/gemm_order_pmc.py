#!/usr/bin/env python3
"""Match the per-dispatch counter rows of `rocprofv3 --pmc <COUNTER> -- python3 tools/gemm_order_sweep.py --manifest M.json` to the
configurations of the manifest (GEMM dispatches in launch order) and print bytes per launch and the ratio to the algorithmic bytes.
usage: gemm_order_pmc.py <counter dir> <manifest.json> <FETCH_SIZE|WRITE_SIZE>"""
import collections, csv, glob, json, os, sys

d, man, counter = sys.argv[1], json.load(open(sys.argv[2])), sys.argv[3]
rows = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "gemm_bf16" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
assert len(rows) == len(man), (len(rows), len(man))
agg = collections.OrderedDict()
for (_, v), c in zip(rows, man):
    key = (c["shape"], c["variant"], c["group_m"], c["col_fast"])
    agg.setdefault(key, []).append((v, c["alg_bytes"]))
mult = 2048.0 if counter == "FETCH_SIZE" else 1024.0      # FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM)
for (shape, variant, gm, cf), vals in agg.items():
    by = sum(v for v, _ in vals[1:]) / max(1, len(vals) - 1) * mult    # first launch of a configuration = warm-up
    print(f"{shape:5s} v{variant:<2d} group_m={gm:<2d} col_fast={cf}: {counter} {by / 1e6:8.1f} MB/launch   (algorithmic total {vals[0][1] / 1e6:.1f} MB)")
