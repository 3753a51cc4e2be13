#!/usr/bin/env python3
"""Micro-benchmark of ucod_gemm_bf16 variants on the backbone's shapes (random data, interleaved rounds in one process)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import native as N, ops

def run(M, Nn, K, epi, variants, rounds=5, iters=10):
    dev = "cuda"
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(Nn, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(Nn, device=dev); sc = torch.ones(Nn, device=dev); resid = torch.randn(M, Nn, device=dev)
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32 if epi == N.EPI_BIAS_SCALE_RESID_F32 else torch.bfloat16)
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            kw = dict(bias=b, variant=v)
            if epi == N.EPI_BIAS_SCALE_RESID_F32: kw.update(scale=sc, resid=resid)
            ops.gemm_bf16(epi, A, W, out, M, Nn, K, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): ops.gemm_bf16(epi, A, W, out, M, Nn, K, **kw)
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / iters * 1e3)
    fl = 2.0 * M * Nn * K
    return {v: (min(t), fl / (min(t) * 1e-6) / 1e12) for v, t in res.items()}

if __name__ == "__main__":
    variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2,3,4,5,6".split(","))]
    M = 32 * 1370
    shapes = [("qkv", M, 2304, 768, N.EPI_BIAS_BF16), ("fc1", M, 3072, 768, N.EPI_BIAS_GELU_BF16), ("proj", M, 768, 768, N.EPI_BIAS_SCALE_RESID_F32),
              ("fc2", M, 768, 3072, N.EPI_BIAS_SCALE_RESID_F32), ("sq4k", 4096, 4096, 4096, N.EPI_BIAS_BF16), ("sq8k", 8192, 8192, 8192, N.EPI_BIAS_BF16),
              ("fc1_nogelu", M, 3072, 768, N.EPI_BIAS_BF16), ("fc2_bf16out", M, 768, 3072, N.EPI_BIAS_BF16)]
    for name, m, n, k, epi in shapes:
        r = run(m, n, k, epi, variants)
        print(f"{name:12s} M={m} N={n} K={k}: " + "  ".join(f"v{v}: {t:7.1f}us {tf:6.1f}TF" for v, (t, tf) in r.items()), flush=True)
