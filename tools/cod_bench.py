#!/usr/bin/env python3
"""COD measures: the device kernels against a host numpy/scipy evaluation of the same measures (the oracle with scipy's
distance transform plugged in, i.e. what the reference's `statistics.step` costs per image) at validation-like sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.ndimage import distance_transform_edt
from ucod_dpl_amd import ops
from oracle import cod_metrics as OM

def scipy_nearest(g):
    d, idx = distance_transform_edt(~g, return_indices=True)
    return d, idx[0], idx[1]
OM.nearest_foreground = scipy_nearest

for B, H, W in ((8, 480, 640), (4, 1024, 1024)):
    g = torch.Generator().manual_seed(H)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    gt = torch.stack([(((yy - H * (0.3 + 0.05 * i)) / (H / 5)) ** 2 + ((xx - W * 0.5) / (W / 4)) ** 2 < 1).float() for i in range(B)])
    pred = ((gt + 0.3 * torch.randn(B, H, W, generator=g)) > 0.5).float()
    pd, gd = pred.cuda(), gt.cuda()
    r = ops.cod_metrics(pd, gd); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): r = ops.cod_metrics(pd, gd)
    torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    ref = OM.image_measures(pred[0].numpy(), gt[0].numpy())
    t_cpu = time.perf_counter() - t0
    got = r[0].cpu().numpy()
    err = max(abs(got[i] - ref[k]) for i, k in enumerate(("mae", "acc", "iou", "sm", "wfm", "adp_em", "adp_fm")))
    print(f"{B} x {H}x{W}: device {t_gpu * 1e3:.2f} ms per batch ({t_gpu / B * 1e3:.3f} ms per image); host numpy/scipy {t_cpu * 1e3:.1f} ms per image; "
          f"ratio {t_cpu / (t_gpu / B):.0f}x; max |diff| of the scalars {err:.2e}")
