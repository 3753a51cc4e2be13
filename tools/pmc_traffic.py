#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the serial bench run.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <state label> > profiles/r01_pmc_traffic.json
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950
(MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, os, sys

CLASSES = {                                  # bench.py kernel class -> substrings of the kernel name (bf16 build | fp16 build with LayerNorm folded, round 5)
    "gemm_bf16_proj_fc2_scale_resid": ("gemm_bf16_mixed_kernel<9,", "gemm_bf16_mixed_kernel<13,"),   # fp16 residual stream (13: + row partials; the f32 form is gemm_bf16_big_kernel<2,)
    "gemm_bf16_fc1_gelu": ("gemm_bf16_mixed_kernel<1,", "gemm_bf16_mixed_kernel<12,"),
    "gemm_bf16_qkv_bias": ("gemm_bf16_mixed_kernel<0,", "gemm_bf16_mixed_kernel<11,"),
    "attention_fwd": ("attn_fwd_v5_kernel",),
    "layernorm": ("layernorm",),
    "row_stats": ("row_stats_h16_kernel",),
}


def mean_per_kernel(d, counter):
    agg = collections.defaultdict(list)
    names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for cls, subs in CLASSES.items():
                if any(sub in r["Kernel_Name"] for sub in subs):
                    agg[cls].append(float(r["Counter_Value"]))
                    names[cls] = r["Kernel_Name"]
                    break
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}, names


fetch, names = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing domains) over `bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline --lora-steps 0 --no-pipeline --streams 1`, MI355X, state '%s'; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                 "launch: FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 (MI355X_MICROARCH.md, HBM section); FETCH_SIZE counts "
                 "L2 fabric requests, Infinity-Cache hits included, so this is an upper bound on HBM traffic; made by tools/pmc_traffic.py" % sys.argv[3],
       "kernels": {}}
for cls in CLASSES:
    if cls in fetch and cls in write:
        fb, wb = 2 * fetch[cls][0] * 1024, write[cls][0] * 1024
        out["kernels"][cls] = {"kernel": names[cls], "launches_sampled": fetch[cls][1], "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb}
print(json.dumps(out, indent=1))
