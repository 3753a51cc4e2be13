#!/usr/bin/env python3
"""How does large-tile GEMM time split into per-K-tile cost and fixed (prologue+epilogue) cost?  Scan K at fixed M,N."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import native as N, ops
M = 32 * 1370
for name, Nn, epi, v in (("qkv-like bf16 out", 2304, N.EPI_BIAS_BF16, 5), ("fc1-like gelu", 3072, N.EPI_BIAS_GELU_BF16, 5),
                         ("proj-like f32 resid", 768, N.EPI_BIAS_SCALE_RESID_F32, 6), ("proj-like bf16", 768, N.EPI_BIAS_BF16, 6)):
    row = []
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(Nn, K, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(Nn, device="cuda"); sc = torch.ones(Nn, device="cuda"); resid = torch.randn(M, Nn, device="cuda")
        out = torch.empty(M, Nn, device="cuda", dtype=torch.float32 if epi == N.EPI_BIAS_SCALE_RESID_F32 else torch.bfloat16)
        kw = dict(bias=b, variant=v)
        if epi == N.EPI_BIAS_SCALE_RESID_F32: kw.update(scale=sc, resid=resid)
        best = 1e9
        for r in range(3):
            ops.gemm_bf16(epi, A, W, out, M, Nn, K, **kw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_bf16(epi, A, W, out, M, Nn, K, **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        row.append(f"K={K}: {best:.0f}us")
    print(name, "v", v, " | ".join(row), flush=True)
