#!/usr/bin/env python3
"""A/B of the parked-tile persistent GEMM experiments (laboratory variants 20: conversion through LDS, 21: transposed accumulator tile +
v_permlane16_swap, no LDS) against the product's mixed-height kernels (13: 256 wide,
14: 192 wide) and plain large-tile kernels (9 / 10) on QKV- and fc1-shaped problems; rows chosen so that 192-wide tiles make whole
rounds (variant 20 has no tall tiles).  Outputs are compared with variant 14 (same tile width, same arithmetic)."""
import statistics
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import ops, native as N  # noqa: E402

VARIANTS = (14, 20, 21, 13, 9, 10)
g = torch.Generator(device="cuda").manual_seed(0)
for name, epi, Nn, K, M in (("qkv", N.EPI_BIAS_BF16, 2304, 768, 32768), ("fc1", N.EPI_BIAS_GELU_BF16, 3072, 768, 32768), ("qkv_c2", N.EPI_BIAS_BF16, 2304, 768, 43840)):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(Nn, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(Nn, device="cuda", generator=g)
    sc = (torch.rand(Nn, device="cuda", generator=g) + 0.5) if epi == N.EPI_BIAS_BF16 else None
    outs, res = {}, {}
    for v in VARIANTS:
        out = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda")
        ops.gemm_bf16(epi, A, W, out, M, Nn, K, bias=b, scale=sc, variant=v)
        torch.cuda.synchronize()
        outs[v] = out
    for rnd in range(5):
        for v in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_bf16(epi, A, W, outs[v], M, Nn, K, bias=b, scale=sc, variant=v)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * Nn * K
    print(name, M, {v: (round(statistics.median(t), 1), round(fl / (statistics.median(t) * 1e-6) / 2.5e15, 3)) for v, t in res.items()},
          "variant 20 == 14:", bool(torch.equal(outs[20], outs[14])), "max|20-13|", float((outs[20].float() - outs[13].float()).abs().max()),
          "max|21-14| / max|14|", float((outs[21].float() - outs[14].float()).abs().max()), float(outs[14].float().abs().max()),
          "elements of 21 that differ from 14:", int((outs[21] != outs[14]).sum()), "of", outs[14].numel())
