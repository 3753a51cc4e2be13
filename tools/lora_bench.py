#!/usr/bin/env python3
"""Backbone-backward (LoRA) mode at the headline size: forward_train + backward of DINOv2 ViT-B/14 @518, per-class kernel times."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucod_dpl_amd import native as N
from ucod_dpl_amd.vit_engine import ViTLoRAEngine
from ucod_dpl_amd.data.utils.feature_extractor import random_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = N.load()
drop = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0   # the reference trains with lora_dropout 0.05 (configs/model/UCOD_DPL.py)
resid = sys.argv[5] if len(sys.argv) > 5 else "auto"      # residual stream of the training pass: auto (fp16 with bf16 operands) / f32
eng = ViTLoRAEngine(random_state_dict("dinov2_vitb14", seed=0), heads=12, device="cuda", lora_dropout=drop, resid=resid)
eng.train_streams = streams
x = torch.randn(B, 3, 518, 518, device="cuda")
dkey = torch.randn(B, 768, 37, 37, device="cuda")
for _ in range(2):
    eng.forward_train(x); eng.backward(dkey)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    eng.forward_train(x)
torch.cuda.synchronize(); t1 = time.time()
for _ in range(steps):
    eng.forward_train(x); eng.backward(dkey)
torch.cuda.synchronize(); t2 = time.time()
fwd = (t1 - t0) / steps * 1e3; both = (t2 - t1) / steps * 1e3
print(f"B={B}: forward_train {fwd:.2f} ms, forward+backward {both:.2f} ms  ({B / both * 1e3:.1f} img/s), workspace {sum(w.numel() for w in eng._tside_ws if w is not None) / 2**30:.2f} GiB in {len(eng._tside)} stream(s)")
eng.train_streams = 1          # exclusive per-kernel durations
eng.forward_train(x); eng.backward(dkey)
torch.cuda.synchronize()
lib.ucod_prof_enable(1)
eng.forward_train(x); eng.backward(dkey)
torch.cuda.synchronize()
lib.ucod_prof_enable(0)
n = lib.ucod_prof_num_classes()
tot = (C.c_double * n)(); cnt = (C.c_longlong * n)()
lib.ucod_prof_collect(tot, cnt)
rows = [(lib.ucod_prof_class_name(i).decode(), cnt[i], tot[i]) for i in range(n) if cnt[i]]
for name, c, t in sorted(rows, key=lambda r: -r[2]):
    print(f"  {name:34s} {c:4d} launches  {t:8.3f} ms total  {t / c * 1e3:8.1f} us avg")
print(f"  sum {sum(r[2] for r in rows):.2f} ms")
