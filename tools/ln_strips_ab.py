#!/usr/bin/env python3
# NOTE (round 6): the environment knobs this tool sets are honoured by the -DUCOD_LAB_KNOBS builds only: `make -C ucod_dpl_amd/csrc knobs`, then run with
#   UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=ucod_dpl_amd/_native/libucod_dpl_knobs.so UCOD_DPL_EXPERIMENT_LIB_F16=ucod_dpl_amd/_native/libucod_dpl_f16_knobs.so
"""fp16-stream LayerNorm (ucod_layernorm_h16, 16-byte strip form) at the C2 and C4 shapes with 1 / 2 / 4 / 8 strips per wave: each setting in
its own process (UCOD_LN_STRIPS is read once).  usage: python tools/ln_strips_ab.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from ucod_dpl_amd import native as N
lib = N.load()
for (M, D) in ((43840, 768), (21920, 1024), (87680, 768)):
    x = torch.randn(M, D, device="cuda").to(torch.float16); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    y = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    run = lambda: N.check(lib.ucod_layernorm_h16(N.ptr(x), N.ptr(g), N.ptr(b), N.ptr(y), M, D, 1e-6, N.stream()), "ln")
    for _ in range(5): run()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    print(f"  {M} x {D}: {best:6.1f} us  {M * D * 4 / best / 1e6:5.2f} TB/s = {M * D * 4 / best / 1e6 / 8:.3f} of 8 TB/s", flush=True)
''' % ROOT
for strips in ("0", "2", "4", "8", "16"):
    print(f"UCOD_LN_STRIPS={strips}" + (" (one strip per wave: the round-3 launch)" if strips == "0" else ""), flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, UCOD_LN_STRIPS=strips), check=True)
