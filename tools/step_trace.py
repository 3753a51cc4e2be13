#!/usr/bin/env python3
"""Order of the launches of the LAST timed step of `bench.py --no-pipeline --streams 1` from a `rocprofv3 --kernel-trace` csv: prints every rocclr fill /
copy with the kernels before and after it, so that each can be attributed to the call that issues it.  usage: step_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
short = lambda n: n.replace("void ", "").replace("ucod::", "").split("(")[0][:60]
# the last step: from the last patch-embedding GEMM back to the one before it is one backbone pass + decoder step
idx = [i for i, n in enumerate(names) if "im2col" in n or "patch_im2col" in n]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
else:
    a, b = 0, len(names)
seq = names[a:b]
print(f"{len(seq)} launches between two consecutive patch im2col launches")
for i, n in enumerate(seq):
    if "rocclr" in n:
        print(f"{i:4d} {short(n):34s} after [{short(seq[i - 1])}]  before [{short(seq[i + 1]) if i + 1 < len(seq) else '-'}]")
