#!/usr/bin/env python3
"""Backbone pass alone (no decoder step) at the bench geometry, 1..3 image-parallel streams: the floor the pipelined step can approach."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd.data.utils.feature_extractor import backbone
bb = backbone.random_init("dinov2_vitb14", seed=0, image_size=518, device="cuda", attn_variant=2)
x = torch.randn(32, 3, 518, 518, device="cuda")
key = torch.empty(32, 768, 37, 37, device="cuda")
for streams in (1, 2, 3):
    bb.engine.streams = streams
    for _ in range(3): bb.engine.forward(x, out=key)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): bb.engine.forward(x, out=key)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"streams {streams}: {dt * 1e3:.3f} ms per pass, {32 / dt:.0f} img/s", flush=True)
