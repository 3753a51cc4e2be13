#!/usr/bin/env python3
"""Why the fp16-operand build is slower: per kernel class, duration and clock of the bf16 and the fp16 build of the SAME serial step.
Reads two rocprofv3 runs (`--pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv`) of
`bench.py --half {bf16,f16} --resid f16 --no-pipeline --streams 1 ...` and prints, per class: launches, mean duration, cycles
(SQ_BUSY_CYCLES / 32 shader engines) and clock = cycles / duration.
usage: clock_pair.py <dir bf16> <dir f16>"""
import collections, csv, glob, os, sys

CLASSES = {"qkv (mixed<0>)": "gemm_bf16_mixed_kernel<0,", "fc1+gelu (mixed<1>)": "gemm_bf16_mixed_kernel<1,", "proj / fc2 (mixed<9>)": "gemm_bf16_mixed_kernel<9,",
           "attention": "attn_fwd_v5_kernel", "layernorm": "layernorm_h16"}


def load(d):
    cyc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "SQ_BUSY_CYCLES":
                continue
            for c, sub in CLASSES.items():
                if sub in r["Kernel_Name"]:
                    cyc[c].append(float(r["Counter_Value"]) / 32.0)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for c, sub in CLASSES.items():
                if sub in r["Kernel_Name"]:
                    dur[c].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return cyc, dur


a, b = load(sys.argv[1]), load(sys.argv[2])
print("# bf16 vs fp16 operands, same kernels, same serial step under rocprofv3 --pmc SQ_BUSY_CYCLES --kernel-trace (a profiled run: compare the two arms, not with un-profiled times)")
print(f"{'class':24s} {'n':>4s} {'bf16 us':>9s} {'f16 us':>9s} {'ratio':>6s} | {'bf16 kcyc':>10s} {'f16 kcyc':>10s} {'ratio':>6s} | {'bf16 GHz':>8s} {'f16 GHz':>8s}")
for c in CLASSES:
    if not (a[1][c] and b[1][c] and a[0][c] and b[0][c]):
        continue
    m = lambda v: sum(v) / len(v)  # noqa: E731
    da, db, ca, cb = m(a[1][c]), m(b[1][c]), m(a[0][c]), m(b[0][c])
    print(f"{c:24s} {len(a[1][c]):4d} {da:9.1f} {db:9.1f} {db / da:6.3f} | {ca / 1e3:10.1f} {cb / 1e3:10.1f} {cb / ca:6.3f} | {ca / da / 1e3:8.2f} {cb / db / 1e3:8.2f}")
