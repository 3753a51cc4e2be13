#!/usr/bin/env python3
# NOTE (round 6): the environment knobs this tool sets are honoured by the -DUCOD_LAB_KNOBS builds only: `make -C ucod_dpl_amd/csrc knobs`, then run with
#   UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=ucod_dpl_amd/_native/libucod_dpl_knobs.so UCOD_DPL_EXPERIMENT_LIB_F16=ucod_dpl_amd/_native/libucod_dpl_f16_knobs.so
"""Tile-order / tile-height / store-policy sweep of the large-tile GEMM on the backbone's four shapes (MI355X).

Every configuration = environment knobs read per launch by csrc/gemm_bf16.hip (UCOD_GEMM_GROUP_M, UCOD_GEMM_COL_FAST) + a variant
number.  Timing: interleaved rounds in one process, min and median over rounds.  With `--manifest FILE` each configuration is
launched a fixed number of times and the launch sequence is written out, so that the per-dispatch rows of a
`rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) pass over this script can be matched to configurations by order
(tools/gemm_order_pmc.py)."""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import native as N, ops

M = 32 * 1370
SHAPES = {"qkv": (M, 2304, 768, N.EPI_BIAS_BF16), "fc1": (M, 3072, 768, N.EPI_BIAS_GELU_BF16),
          "proj": (M, 768, 768, N.EPI_BIAS_SCALE_RESID_F32), "fc2": (M, 768, 3072, N.EPI_BIAS_SCALE_RESID_F32)}
ALG = {k: (m * kk * 2 + n * kk * 2 + m * n * (8 if e == N.EPI_BIAS_SCALE_RESID_F32 else 2)) for k, (m, n, kk, e) in SHAPES.items()}


def configs(mode="order"):
    out = []
    if mode == "order":
        for variant in (9, 13, 10):
            for gm, cf in ((8, 0), (4, 0), (16, 0), (2, 1), (4, 1), (8, 1), (0, 1)):
                out.append(dict(variant=variant, group_m=gm, col_fast=cf))
    else:                                    # store policy / next-tile prefetch of the mixed-height kernel, in one process
        for pf in (0, 1):
            for aux in (0, 2, 16):
                out.append(dict(variant=13, group_m=8, col_fast=0, prefetch=pf, aux=aux))
        out.append(dict(variant=0, group_m=8, col_fast=0))
        out.append(dict(variant=9, group_m=8, col_fast=0))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--shapes", default="qkv,fc1,proj,fc2")
    ap.add_argument("--mode", default="order", choices=["order", "policy"])
    a = ap.parse_args()
    dev = "cuda"
    manifest = []
    for name in a.shapes.split(","):
        m, n, k, epi = SHAPES[name]
        A = torch.randn(m, k, device=dev).to(torch.bfloat16)
        W = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        b, sc = torch.randn(n, device=dev), torch.ones(n, device=dev)
        f32 = epi == N.EPI_BIAS_SCALE_RESID_F32
        out = torch.zeros(m, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        cfgs = configs(a.mode)
        times = {i: [] for i in range(len(cfgs))}

        def launch(c):
            os.environ["UCOD_GEMM_GROUP_M"] = str(c["group_m"])
            os.environ["UCOD_GEMM_COL_FAST"] = str(c["col_fast"])
            os.environ["UCOD_GEMM_PREFETCH"] = str(c.get("prefetch", 0))
            os.environ["UCOD_GEMM_ST_AUX"] = str(c.get("aux", 0))
            kw = dict(bias=b, variant=c["variant"])
            if f32:
                kw.update(scale=sc, resid=out)
            ops.gemm_bf16(epi, A, W, out, m, n, k, **kw)

        if a.manifest:
            for i, c in enumerate(cfgs):
                for _ in range(3):
                    launch(c)
                    manifest.append(dict(shape=name, alg_bytes=ALG[name], **c))
            torch.cuda.synchronize()
            continue
        for r in range(a.rounds):
            for i, c in enumerate(cfgs):
                launch(c)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    launch(c)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / a.iters * 1e3)
        fl = 2.0 * m * n * k
        for i, c in enumerate(cfgs):
            t = times[i]
            print(f"{name:5s} v{c['variant']:<2d} group_m={c['group_m']:<2d} col_fast={c['col_fast']} pf={c.get('prefetch', 0)} aux={c.get('aux', 0):<2d}: min {min(t):7.1f} us  med {statistics.median(t):7.1f} us  "
                  f"{fl / (min(t) * 1e-6) / 1e12:6.1f} TF/s ({fl / (min(t) * 1e-6) / 2.5e15:.3f} of peak)", flush=True)
    if a.manifest:
        json.dump(manifest, open(a.manifest, "w"))


if __name__ == "__main__":
    main()
