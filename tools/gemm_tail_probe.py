#!/usr/bin/env python3
"""How much of a backbone GEMM is the last partial round of tiles?  Times each shape at M = 43840 (the bench's 32 x 1370 rows:
tile counts just above a multiple of the 512 co-resident slots) and at M = 43520 (170 row tiles: just below)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucod_dpl_amd import native as N
from tools.gemm_bench import run

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for name, n, k, epi in (("qkv", 2304, 768, N.EPI_BIAS_BF16), ("fc1", 3072, 768, N.EPI_BIAS_GELU_BF16), ("proj", 768, 768, N.EPI_BIAS_SCALE_RESID_F32),
                        ("fc2", 768, 3072, N.EPI_BIAS_SCALE_RESID_F32)):
    for m in (43520, 43776, 43840):
        t, tf = run(m, n, k, epi, [variant])[variant]
        tiles = -(-m // 256) * (n // 256)
        print(f"{name:5s} M={m} tiles={tiles:5d} ({tiles / 512:.3f} rounds): {t:7.1f} us {tf:7.1f} TF  {t / m * 1e3:.3f} ns/row", flush=True)
