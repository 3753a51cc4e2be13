#!/usr/bin/env python3
"""Attention-only benchmark at the C2 shape (B=32, 12 heads, N=1370)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import ops
B, tok, heads = 32, 1370, 12
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
qkv_f = torch.randn(B * tok, 3 * heads * 64, device="cuda")
qkv = qkv_f.to(torch.bfloat16)
qkv_f[:, :heads * 64] *= 0.125 * 1.4426950408889634
qkv_pre = qkv_f.to(torch.bfloat16)          # Q pre-scaled, as the QKV epilogue hands it to the v2 kernel
fl = 4.0 * B * heads * tok * tok * 64
for r in range(3):
    for v in variants:
        sc = 0.0 if v >= 2 else 0.125
        x = qkv_pre if v >= 2 else qkv
        ops.attention(x, B, tok, heads, scale=sc, variant=v); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): ops.attention(x, B, tok, heads, scale=sc, variant=v)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e3
        print(f"attn v{v}: {t:.1f} us  {fl / (t * 1e-6) / 1e12:.1f} TF", flush=True)
