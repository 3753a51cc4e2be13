import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import ops
DEV = "cuda"
C, H, Nout = 768, 37, 256
for B in (4, 16, 32):
    x = torch.randn(B, C, H, H, device=DEV); W = torch.randn(Nout, C, device=DEV) / 27; b = torch.randn(Nout, device=DEV)
    out = []
    for ex in (False, True):
        for _ in range(3): ops.dba_project(x, W, b, exact=ex)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.dba_project(x, W, b, exact=ex)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"B={B}: split {out[0]:.1f} us, exact {out[1]:.1f} us   ({15 * B} workgroups)")
