#!/usr/bin/env python3
"""Per-kernel times of the split-operand pass at BASELINE configs[1]'s shape (32 x 1370 tokens, D = 768): `python tools/split_bench.py [--terms 2] [--batch 32]`.
HIP events around `iters` back-to-back launches of each piece, after a warm-up."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucod_dpl_amd import native as N, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--terms", type=int, default=2)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev, T, B, tok, heads, D, F = "cuda", a.terms, a.batch, 1370, 12, 768, 3072
M, P = B * tok, ops.split_products(a.terms)
lib = N.load()
g = torch.Generator().manual_seed(0)


def timed(name, fn, flops=None, nbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    extra = (f"  {flops / us / 1e6:8.1f} TF/s algorithmic ({flops * P / us / 1e6:7.1f} issued)" if flops else "") + (f"  {nbytes / us / 1e3:8.1f} GB/s" if nbytes else "")
    print(f"{name:34s} {us:9.1f} us{extra}", flush=True)


x = torch.randn(M, D, generator=g).to(dev)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
qkv = torch.randn(M, 3 * D, generator=g).to(dev)
f1 = torch.randn(M, F, generator=g).to(dev)
need = lib.ucod_attention_split_operand_bytes(B, tok, heads, T)
opnd = torch.empty(need, dtype=torch.uint8, device=dev)
aout = torch.empty(M, P * D, dtype=torch.bfloat16, device=dev)
hs = torch.empty(M, P * D, dtype=torch.bfloat16, device=dev)
gs = torch.empty(M, P * F, dtype=torch.bfloat16, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
print(f"# split-operand pieces, terms = {T} (P = {P}), {B} x {tok} tokens, D = {D}")
timed("layernorm_split", lambda: N.check(lib.ucod_layernorm_split(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), hs.data_ptr(), M, D, 1e-6, T, 0, st()), "ln"), nbytes=M * D * (4 + 2 * P))
timed("qkv_split", lambda: N.check(lib.ucod_qkv_split(qkv.data_ptr(), opnd.data_ptr(), B, tok, heads, T, 0.18, st()), "qs"), nbytes=M * 3 * D * 4 + need)
timed("attention_split_fwd", lambda: N.check(lib.ucod_attention_split_fwd(opnd.data_ptr(), aout.data_ptr(), B, tok, heads, T, st()), "att"), flops=4.0 * B * heads * tok * tok * 64)
timed("split_rows(gelu) fc1 out", lambda: N.check(lib.ucod_split_rows(f1.data_ptr(), F, gs.data_ptr(), M, F, T, 0, 1, 1.0, st()), "sg"), nbytes=M * F * (4 + 2 * P))
wq = ops.split_rows(torch.randn(3 * D, D, generator=g).to(dev) * 0.02, T, 1)
wp = ops.split_rows(torch.randn(D, D, generator=g).to(dev) * 0.02, T, 1)
w1 = ops.split_rows(torch.randn(F, D, generator=g).to(dev) * 0.02, T, 1)
w2 = ops.split_rows(torch.randn(D, F, generator=g).to(dev) * 0.02, T, 1)
bq, bp, b1 = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(F, device=dev)
ls = torch.ones(D, device=dev)
oq = torch.empty(M, 3 * D, device=dev)
timed("GEMM qkv  (BIAS_F32)", lambda: ops.gemm_bf16(N.EPI_BIAS_F32, hs, wq, oq, M, 3 * D, P * D, bias=bq), flops=2.0 * M * 3 * D * D)
timed("GEMM fc1  (BIAS_F32)", lambda: ops.gemm_bf16(N.EPI_BIAS_F32, hs, w1, f1, M, F, P * D, bias=b1), flops=2.0 * M * F * D)
if T == 2:
    timed("GEMM fc1  (GELU_SPLIT2: fused)", lambda: ops.gemm_bf16(N.EPI_BIAS_GELU_SPLIT2, hs, w1, gs, M, F, P * D, bias=b1), flops=2.0 * M * F * D)
timed("GEMM proj (SCALE_RESID_F32)", lambda: ops.gemm_bf16(N.EPI_BIAS_SCALE_RESID_F32, aout, wp, x, M, D, P * D, bias=bp, scale=ls, resid=x), flops=2.0 * M * D * D)
timed("GEMM fc2  (SCALE_RESID_F32)", lambda: ops.gemm_bf16(N.EPI_BIAS_SCALE_RESID_F32, gs, w2, x, M, D, P * F, bias=bp, scale=ls, resid=x), flops=2.0 * M * D * F)
