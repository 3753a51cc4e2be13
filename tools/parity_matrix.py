#!/usr/bin/env python3
"""Full-size parity of the backbone + decoder against the f32 oracle for every (operand type, residual-stream type) combination:
4 images at 518x518, DINOv2 ViT-B/14, random-init weights (the bench's parity leg as a matrix).  Run on the MI355X box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vit as OV, decoder as OD
from oracle.resize import torch_bilinear
from ucod_dpl_amd import ops
from ucod_dpl_amd.vit_engine import ViTEngine
from ucod_dpl_amd.data.utils.feature_extractor import random_state_dict

n, D, heads = 4, 768, 12
torch.manual_seed(0)
sd = random_state_dict("dinov2_vitb14", 0, 518)
img = torch.randn(n, 3, 518, 518)
gen = torch.Generator().manual_seed(42)
dec = OD.init_params(D, gen)
with torch.no_grad():
    _, key = OV.dinov2_forward(img, sd, heads=heads, patch=14, eps=1e-6, full_last_layer=False)
    fg_ref, _, _ = OD.rev_decoder_forward(torch_bilinear(key, 68, 68), dec, orth="gram")
dev = torch.device("cuda")
emb = dec["learnable_embedding"].reshape(128).to(dev)
hw = torch.cat((dec["conv_out_fg.weight"].reshape(64), dec["conv_out_bg.weight"].reshape(64))).to(dev)
hb = torch.cat((dec["conv_out_fg.bias"], dec["conv_out_bg.bias"])).to(dev)
for half in ("bf16", "f16"):
    for resid in ("f32", "f16"):
        eng = ViTEngine(sd, heads=heads, eps=1e-6, device=dev, attn_variant=2, half=half, resid=resid)
        kd = eng(img.to(dev))
        d = ops.bilinear_resize(ops.dba_project(kd, dec["decoupling.weight"].reshape(128, D).to(dev), dec["decoupling.bias"].to(dev)).view(n, 128, 37, 37), 68, 68).view(n, 128, 68 * 68)
        fd = ops.dba_heads(d, 0, emb, ops.dba_colnorm(d, 0, emb), hw, hb, want_bg=False)[0].view(n, 1, 68, 68).cpu()
        kd = kd.cpu()
        print(f"operands {half:4s} residual {resid}: key rel-L2 {float((kd - key).norm() / key.norm()):.3e}  logit max-abs {float((fd - fg_ref).abs().max()):.3e}  "
              f"rel-L2 {float((fd - fg_ref).norm() / fg_ref.norm()):.3e}  flipped {float(((fd > 0) != (fg_ref > 0)).float().mean()):.1e}", flush=True)
