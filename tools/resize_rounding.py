#!/usr/bin/env python3
"""Which rounding pattern of the bilinear four-tap blend reproduces ATen's CPU F.interpolate bit for bit?  (CPU only: numpy
float32 with fma emulated through float64.)  Per axis w0*a + w1*b can be fma(b, w1, rn(a*w0)) [0], fma(a, w0, rn(b*w1)) [1] or
rn(a*w0) + rn(b*w1) [2]; x inside y.  Prints the fraction of exactly equal outputs per (inner, outer) pattern and geometry:
(1, 1) is exact on 37 -> 68 and 16 -> 68 -- the pattern pinned in csrc/elementwise.hip lerp2()."""
import itertools
import numpy as np
import torch

f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
def mul(a, b): return (a * b).astype(f32)
def src_index(o, scale, n):
    s = np.maximum(fma(np.full(o.shape, scale, dtype=f32), o.astype(f32) + f32(0.5), np.full(o.shape, -0.5, dtype=f32)), f32(0))
    i0 = np.minimum(s.astype(np.int32), n - 1)
    return i0, i0 + (i0 < n - 1), (s - i0.astype(f32)).astype(f32)
def comb(a, wa, b, wb, mode):
    return fma(b, wb, mul(a, wa)) if mode == 0 else fma(a, wa, mul(b, wb)) if mode == 1 else (mul(a, wa) + mul(b, wb)).astype(f32)

torch.manual_seed(0)
for ih, oh, planes in ((37, 68, 16), (16, 68, 8), (14, 28, 64)):
    x = torch.randn(1, planes, ih, ih)
    ref = torch.nn.functional.interpolate(x, size=(oh, oh), mode="bilinear", align_corners=False)[0].numpy()
    xn, o = x[0].numpy(), np.arange(oh)
    y0, y1, ly = src_index(o, f32(ih) / f32(oh), ih)
    x0, x1, lx = y0, y1, ly
    v00, v01, v10, v11 = xn[:, y0][:, :, x0], xn[:, y0][:, :, x1], xn[:, y1][:, :, x0], xn[:, y1][:, :, x1]
    LX, LY = np.broadcast_to(lx[None, None, :], v00.shape).astype(f32), np.broadcast_to(ly[None, :, None], v00.shape).astype(f32)
    WX, WY = (f32(1) - LX).astype(f32), (f32(1) - LY).astype(f32)
    for mi, mo in itertools.product(range(3), range(3)):
        out = comb(comb(v00, WX, v01, LX, mi), WY, comb(v10, WX, v11, LX, mi), LY, mo)
        print(f"{ih:3d} -> {oh:3d}  inner {mi} outer {mo}: {(out == ref).mean():.4f}")
