#!/usr/bin/env python3
# NOTE (round 6): the environment knobs this tool sets are honoured by the -DUCOD_LAB_KNOBS builds only: `make -C ucod_dpl_amd/csrc knobs`, then run with
#   UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=ucod_dpl_amd/_native/libucod_dpl_knobs.so UCOD_DPL_EXPERIMENT_LIB_F16=ucod_dpl_amd/_native/libucod_dpl_f16_knobs.so
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

# fp16 residual stream -> operand halves (the form the backbone pass uses on >= 4096 rows); UCOD_LN_NO_STRIP=1 selects the 8-byte kernel
from ucod_dpl_amd import native as N
lib = N.load()
xh = x.to(torch.float16); y = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
def run(): N.check(lib.ucod_layernorm_h16(N.ptr(xh), N.ptr(g), N.ptr(b), N.ptr(y), M, D, 1e-6, N.stream()), "ln_h16")
for _ in range(3): run()
torch.cuda.synchronize()
best = 1e9
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
print(f"layernorm_h16 {M}x{D} ({'8-byte' if os.environ.get('UCOD_LN_NO_STRIP') else '16-byte strip'} form): {best:.1f} us  {M * D * 4 / best / 1e3:.0f} GB/s")
