#!/usr/bin/env python3
"""LayerNorm micro-benchmark at the backbone's shape (43840 x 768 f32 -> bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import ops
M, D = 32 * 1370, 768
x = torch.randn(M, D, device="cuda"); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
for _ in range(3): ops.layernorm(x, g, b, 1e-6)
torch.cuda.synchronize()
best = 1e9
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layernorm(x, g, b, 1e-6)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
print(f"layernorm {M}x{D}: {best:.1f} us  {M * D * 6 / best / 1e3:.0f} GB/s")
