#!/usr/bin/env python3
"""Per-class kernel times of the discriminator phase (row A8) at the bench geometry."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ucod_dpl_amd import native
from ucod_dpl_amd.engine.runner import StandardRunner, TrainLoop
lib = native.load()
cfg = bench.make_cfg(68, 768)
runner = StandardRunner(cfg); loop = TrainLoop(cfg, runner)
g = torch.Generator().manual_seed(1)
key = torch.randn(32, 768, 37, 37, generator=g).cuda(); pl = (torch.rand(32, 1, 16, 16, generator=g) > 0.7).float().cuda()
for _ in range(3): loop._discriminator_batch((pl, key))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loop._discriminator_batch((pl, key))
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
lib.ucod_prof_enable(1)
for _ in range(5): loop._discriminator_batch((pl, key))
torch.cuda.synchronize(); lib.ucod_prof_enable(0)
n = lib.ucod_prof_num_classes(); tot = (C.c_double * n)(); cnt = (C.c_longlong * n)(); lib.ucod_prof_collect(tot, cnt)
for i in range(n):
    if cnt[i]: print(f"  {lib.ucod_prof_class_name(i).decode():28s} {cnt[i] / 5:5.1f} launches/step {tot[i] / 5:8.3f} ms/step")
