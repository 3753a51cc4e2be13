#!/usr/bin/env python3
"""Resize kernels of the decoder step at the bench geometry: d (8192 planes 37x37 -> 68x68) and the adjoint of gd (4096 planes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import ops
x = torch.randn(32 * 256, 37, 37, device="cuda")
g = torch.randn(32 * 128, 68, 68, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = t(lambda: ops.bilinear_resize(x, 68, 68))
b = t(lambda: ops.bilinear_resize_adjoint(g, 37, 37))
print(f"resize 37->68, 8192 planes: {a:.1f} us  ({(x.numel() + x.shape[0] * 4624) * 4 / a / 1e6:.2f} TB/s)")
print(f"adjoint 68->37, 4096 planes: {b:.1f} us  ({(g.numel() + g.shape[0] * 1369) * 4 / b / 1e6:.2f} TB/s)")
