#!/usr/bin/env python3
"""A/B of the large-tile GEMM on 32 x 32 x 16 MFMAs (laboratory variant 22) against the same kernel on 16 x 16 x 32 (product variant 9: plain large
tiles, 256 wide) and the product default (13: mixed-height), on the four backbone shapes; whole rounds (M = 32768) and the C2 row count (43840).
Outputs are compared with variant 9 (same tile, same K order per output: the f32 sums differ only by the MFMA shape's internal order)."""
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import ops, native as N  # noqa: E402

VARIANTS = (9, 22, 13)
g = torch.Generator(device="cuda").manual_seed(0)
for name, epi, Nn, K, M in (("qkv", N.EPI_BIAS_BF16, 2304, 768, 32768), ("fc1", N.EPI_BIAS_GELU_BF16, 3072, 768, 32768), ("proj", N.EPI_BIAS_BF16, 768, 768, 32768),
                            ("fc2", N.EPI_BIAS_BF16, 768, 3072, 32768), ("qkv_c2", N.EPI_BIAS_BF16, 2304, 768, 43840), ("fc1_c2", N.EPI_BIAS_GELU_BF16, 3072, 768, 43840)):
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
    ref = (A.float() @ W.float().t() + b)
    ref = torch.nn.functional.gelu(ref) if epi == N.EPI_BIAS_GELU_BF16 else (ref * sc if sc is not None else ref)
    err = {v: float((outs[v].float() - ref).abs().max() / ref.abs().max()) for v in VARIANTS}
    print(name, M, {v: (round(statistics.median(t), 1), round(fl / (statistics.median(t) * 1e-6) / 2.5e15, 3)) for v, t in res.items()},
          "max|22-9|", float((outs[22].float() - outs[9].float()).abs().max()), "differing", int((outs[22] != outs[9]).sum()), "of", outs[9].numel(),
          "rel err vs f32 matmul", {v: round(e, 5) for v, e in err.items()})
