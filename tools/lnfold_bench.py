#!/usr/bin/env python3
# NOTE (round 6): the environment knobs this tool sets are honoured by the -DUCOD_LAB_KNOBS builds only: `make -C ucod_dpl_amd/csrc knobs`, then run with
#   UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=ucod_dpl_amd/_native/libucod_dpl_knobs.so UCOD_DPL_EXPERIMENT_LIB_F16=ucod_dpl_amd/_native/libucod_dpl_f16_knobs.so
"""Isolated, interleaved timing of the LayerNorm-folded GEMMs against what they replace (BASELINE configs[1] shapes, fp16-operand build):
   LayerNorm kernel, row-statistics kernel, QKV / fc1 unfolded, folded with `stats`, folded with row partials, out-projection with and without
   the partial-sum epilogue.  Optional ablation libraries (make variant_f16 NAME=.. DEFS=-DUCOD_FOLD_ABL=n) are timed beside the product library.
usage: lnfold_bench.py [images=32] [reps=30] [lib.so ...]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ucod_dpl_amd import native as N, ops  # noqa: E402


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in N.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    libs = {"product": N.load("f16")}
    for pth in sys.argv[3:]:
        libs[os.path.basename(pth).replace("libucod_dpl_", "").replace(".so", "")] = load(pth)
    dev = "cuda"
    D, F, tok = 768, 3072, 1370
    M = B * tok
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, D, generator=g)).to(torch.float16).to(dev)
    gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    wq, bq = (torch.randn(3 * D, D, generator=g) * 0.03).to(dev), torch.zeros(3 * D, device=dev)
    w1, b1 = (torch.randn(F, D, generator=g) * 0.03).to(dev), torch.zeros(F, device=dev)
    wo = (torch.randn(D, D, generator=g) * 0.03).to(torch.float16).to(dev)
    fq, f1 = ops.fold_layernorm_linear(gamma, beta, wq, bq), ops.fold_layernorm_linear(gamma, beta, w1, b1)
    wq16, w116 = wq.to(torch.float16), w1.to(torch.float16)
    h = torch.empty(M, D, dtype=torch.float16, device=dev)
    qkv = torch.empty(M, 3 * D, dtype=torch.float16, device=dev)
    hid = torch.empty(M, F, dtype=torch.float16, device=dev)
    att = (torch.randn(M, D, generator=g) * 0.5).to(torch.float16).to(dev)
    stats = torch.empty(M, 2, dtype=torch.float32, device=dev)
    part = torch.empty(M, D // 64, 2, dtype=torch.float32, device=dev)
    ones = torch.ones(D, device=dev)
    qs = torch.ones(3 * D, device=dev)
    st = N.stream()
    p = N.ptr

    def cases(lib):
        return {
            "layernorm_h16": lambda: lib.ucod_layernorm_h16(p(x), p(gamma), p(beta), p(h), M, D, 1e-6, st),
            "row_stats_h16": lambda: lib.ucod_row_stats_h16(p(x), p(stats), M, D, 1e-6, st),
            "qkv_unfolded": lambda: lib.ucod_gemm_bf16(N.EPI_BIAS_BF16, p(h), p(wq16), p(qkv), M, 3 * D, D, p(bq), p(qs), None, None, tok, 0, st),
            "qkv_fold_stats": lambda: lib.ucod_gemm_lnfold(N.EPI_LNFOLD_BIAS_BF16, p(x), p(fq[0]), p(qkv), M, 3 * D, D, p(fq[1]), p(fq[2]), p(stats), None, 0, 1e-6, None, 0, st),
            "qkv_fold_part": lambda: lib.ucod_gemm_lnfold(N.EPI_LNFOLD_BIAS_BF16, p(x), p(fq[0]), p(qkv), M, 3 * D, D, p(fq[1]), p(fq[2]), None, p(part), D // 64, 1e-6, None, 0, st),
            "fc1_unfolded": lambda: lib.ucod_gemm_bf16(N.EPI_BIAS_GELU_BF16, p(h), p(w116), p(hid), M, F, D, p(b1), None, None, None, tok, 0, st),
            "fc1_fold_stats": lambda: lib.ucod_gemm_lnfold(N.EPI_LNFOLD_GELU_BF16, p(x), p(f1[0]), p(hid), M, F, D, p(f1[1]), p(f1[2]), p(stats), None, 0, 1e-6, None, 0, st),
            "fc1_fold_part": lambda: lib.ucod_gemm_lnfold(N.EPI_LNFOLD_GELU_BF16, p(x), p(f1[0]), p(hid), M, F, D, p(f1[1]), p(f1[2]), None, p(part), D // 64, 1e-6, None, 0, st),
            "proj_resid_h16": lambda: lib.ucod_gemm_bf16(N.EPI_BIAS_SCALE_RESID_H16, p(att), p(wo), p(x), M, D, D, p(gamma), p(ones), p(x), None, tok, 0, st),
            "proj_resid_h16_stats": lambda: lib.ucod_gemm_bf16_stats(N.EPI_BIAS_SCALE_RESID_H16_STATS, p(att), p(wo), p(x), M, D, D, p(gamma), p(ones), p(x), None, tok, p(part), D // 64, st),
        }

    table = {name: cases(lib) for name, lib in libs.items()}
    product = table["product"]
    product["row_stats_h16"]()
    product["proj_resid_h16_stats"]()
    torch.cuda.synchronize()
    times = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rnd in range(reps + 3):
        for lname, cs in table.items():
            for cname, fn in cs.items():
                if lname != "product" and "fold" not in cname:
                    continue
                x.normal_()                                   # (keeps the in-place residual cases bounded; also flushes nothing: same for every case)
                ev[0].record()
                rc = fn()
                ev[1].record()
                assert rc == 0, (lname, cname, rc)
                torch.cuda.synchronize()
                if rnd >= 3:
                    times.setdefault((lname, cname), []).append(ev[0].elapsed_time(ev[1]) * 1e3)
    for (lname, cname), v in times.items():
        v.sort()
        print(f"{lname:14s} {cname:22s} median {v[len(v) // 2]:8.1f} us   min {v[0]:8.1f}")


if __name__ == "__main__":
    main()
