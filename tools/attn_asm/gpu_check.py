#!/usr/bin/env python3
"""First-contact check of the assembly attention kernel on a GPU: small shapes against an f64 softmax and against attn_fwd_v5_kernel,
the forced-rescale inputs, the LSE output, then the C2 shape.  Each stage prints one line; run under `timeout`."""
import math
import os
import sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ucod_dpl_amd import native as N, ops  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "small"
AV = int(os.environ.get("ATTN_VARIANT", "64"))
DEV = "cuda"


def ref_attn(qkv, B, tok, heads):
    D = heads * 64
    x = qkv.double().reshape(B, tok, 3, heads, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    s = q @ k.transpose(2, 3)
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    l = p.sum(-1, keepdim=True)
    o = ((p / l) @ v).transpose(1, 2).reshape(B * tok, D)
    return o, (m + torch.log2(l))[..., 0]


def run(qkv, B, tok, heads, variant):
    if variant in (64, 32):                                  # the assembly kernels: laboratory library since round 5 (make -C ucod_dpl_amd/csrc variants)
        return ops.attention_asm(qkv.to(DEV), B, tok, heads, form=0 if variant == 64 else 1).float().cpu()
    return ops.attention(qkv.to(DEV), B, tok, heads, scale=0.0, variant=variant).float().cpu()


if stage == "small":
    for (B, tok, heads) in [(1, 200, 1), (1, 129, 2), (2, 300, 2), (1, 1370, 3), (3, 520, 1), (2, 1370, 12)]:
        g = torch.Generator().manual_seed(tok + heads)
        D = heads * 64
        qkv = torch.randn(B * tok, 3 * D, generator=g) * 1.5
        qkv[:, :D] *= 0.125 * math.log2(math.e)
        qkv = qkv.to(torch.bfloat16)
        ref, _ = ref_attn(qkv, B, tok, heads)
        o3 = run(qkv, B, tok, heads, AV)
        o5 = run(qkv, B, tok, heads, 5)
        e3, e5 = (o3.double() - ref).abs().max().item(), (o5.double() - ref).abs().max().item()
        r3 = ((o3.double() - ref).norm() / ref.norm()).item()
        again = run(qkv, B, tok, heads, AV)
        print(f"shape {(B, tok, heads)}: asm max|err| {e3:.4g} rel-L2 {r3:.3g}  (v5 {e5:.4g})  repeat-bitwise {torch.equal(o3, again)}  nan {torch.isnan(o3).any().item()}", flush=True)
elif stage == "branches":
    D, tok = 64, 400
    c = 0.125 * math.log2(math.e)
    g = torch.Generator().manual_seed(21)
    base = torch.randn(tok, 3 * D, generator=g) * 0.3
    a = base.clone(); a[5, :D] = 2.0; a[333, D:2 * D] = 6.0
    b_ = base.clone(); b_[:, D:2 * D] += torch.linspace(0, 1.2, tok).view(-1, 1) * 0.5; b_[:, :D] = 0.5
    c_ = base.clone(); c_[:64, D:2 * D] += 3.0; c_[:, :D] = 1.0
    for name, x in (("spike", a), ("creep", b_), ("first-dominates", c_)):
        x = x.clone(); x[:, :D] *= c
        x = x.to(torch.bfloat16)
        ref, _ = ref_attn(x, 1, tok, 1)
        o3 = run(x, 1, tok, 1, AV)
        print(f"{name}: asm max|err| {(o3.double() - ref).abs().max().item():.4g} nan {torch.isnan(o3).any().item()}", flush=True)
elif stage == "lse":
    B, tok, heads = 2, 1370, 3
    g = torch.Generator().manual_seed(5)
    D = heads * 64
    qkv = torch.randn(B * tok, 3 * D, generator=g) * 1.5
    qkv[:, :D] *= 0.125 * math.log2(math.e)
    qkv = qkv.to(torch.bfloat16)
    ref, lse_ref = ref_attn(qkv, B, tok, heads)
    out, lse = ops.attention_asm(qkv.to(DEV), B, tok, heads, form=0 if AV == 64 else 1, want_lse=True)
    rc = 0
    torch.cuda.synchronize()
    print(f"lse path rc {rc}: out max|err| {(out.float().cpu().double() - ref).abs().max().item():.4g}  lse max|err| {(lse.cpu().double() - lse_ref).abs().max().item():.4g}", flush=True)
