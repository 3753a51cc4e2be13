#!/usr/bin/env python3
"""Timings of the "next" rows of SURVEY.md 8f on one MI355X: N1 feature-cache pass, N2 device Look-Twice tail, N3 pseudo-label
generator (N4's COD measures: tools/cod_bench.py).  Synthetic inputs of the real sizes; results are printed, not asserted."""
import os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ucod_dpl_amd import native as N
from ucod_dpl_amd.data.utils.feature_extractor import backbone
from ucod_dpl_amd.data.datasets import MultiCacheManager, build_feature_cache
from ucod_dpl_amd.generate_pseudo_label import PseudoLabelGenerator

dev = "cuda"
def sync_time(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# ---- N1: the cache-building pass (base_dataset.py:124-145): backbone over images, items streamed to features_cache/.../data_i.pkl
bb = backbone.random_init("dinov2_vitb14", seed=0, image_size=518, device=dev, attn_variant=2)
imgs = [torch.randn(3, 518, 518) for _ in range(128)]
x = torch.stack(imgs[:32]).to(dev)
tb = sync_time(lambda: bb(x), 3)                                      # also the warm-up of the pass
t0 = time.perf_counter(); k = bb(x)[1].to("cpu"); torch.cuda.synchronize(); td2h = time.perf_counter() - t0
tmp = tempfile.mkdtemp(prefix="ucod_cache_")
try:
    mc = MultiCacheManager(tmp, "dinov2_vitb14", "train", "SYNTH")
    eq = bb.with_precision("f32eq")                                    # what the pass uses by default since round 6 (the reference's fp32: base_dataset.py:124-138)
    teq = sync_time(lambda: eq(x), 2)
    t0 = time.perf_counter(); n = build_feature_cache(imgs, bb, mc.get_features_cache(), batch_size=32, device=dev); t1 = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
    tmp2 = tempfile.mkdtemp(prefix="ucod_cache16_")
    t0 = time.perf_counter(); build_feature_cache(imgs, bb, MultiCacheManager(tmp2, "dinov2_vitb14", "train", "SYNTH").get_features_cache(), batch_size=32, device=dev, precision=None); t16 = time.perf_counter() - t0
    shutil.rmtree(tmp2, ignore_errors=True)
    print(f"N1 (round 6) the pass runs the f32-equivalent split-operand engine by default: {32 / teq:.0f} img/s for the pass alone ({teq * 1e3:.1f} ms per 32 images); end to end "
          f"{n / t1:.0f} img/s against {n / t16:.0f} with the fp16 engine (precision=None): the on-disk format, not the device, sets the pace")
    print(f"N1 feature cache: {n} images at 518x518 in {t1:.2f} s = {n / t1:.0f} img/s end to end, {size / 1e6:.0f} MB of per-item pickles written "
          f"({size / 1e6 / t1:.0f} MB/s: host-side stacking, D2H and file I/O bound -- one 32-image batch: pass {tb * 1e3:.1f} ms, pass + copy of the "
          f"134 MB key maps to the host {td2h * 1e3:.0f} ms); the batched backbone pass alone {32 / tb:.0f} img/s (the reference runs it one image per call)")
finally:
    shutil.rmtree(tmp, ignore_errors=True)

# ---- N3: pseudo-label generator (generate_pseudo_label.py:30-94) at its 224x224 / 16x16 geometry
bb224 = backbone.random_init("dinov2_vitb14", seed=0, image_size=224, device=dev, attn_variant=2)
gen = PseudoLabelGenerator(bb224)
x224 = torch.randn(32, 3, 224, 224, device=dev)
tr = sync_time(lambda: gen.raw_masks(x224), 5)
t0 = time.perf_counter(); masks = gen.generate_masks(x224); tg = time.perf_counter() - t0
print(f"N3 pseudo labels: backbone + CLS attention row + seed similarity + threshold for 32 images at 224x224: {tr * 1e3:.2f} ms ({32 / tr:.0f} img/s) on the device; "
      f"with the host small-component post-process {tg * 1e3:.1f} ms per batch")

# ---- N2: device connected components + bounding boxes, and bicubic resize + paste
lib = N.load()
H = W = 518
yy, xx = np.mgrid[0:H, 0:W]
mask = np.zeros((H, W), np.uint8)
rng = np.random.default_rng(0)
for _ in range(6):
    cy, cx, ry, rx = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(10, 80), rng.uniform(10, 80)
    mask[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1] = 255
md = torch.from_numpy(mask).to(dev)
ws = torch.empty(lib.ucod_ccl8_workspace_bytes(H, W), dtype=torch.uint8, device=dev)
table = torch.empty(4096, 6, dtype=torch.int32, device=dev); count = torch.zeros(1, dtype=torch.int32, device=dev)
def ccl():
    count.zero_()
    N.check(lib.ucod_ccl8_components(N.ptr(md), H, W, N.ptr(table), 4096, N.ptr(count), N.ptr(ws), ws.numel(), N.stream()), "ccl")
tc = sync_time(ccl, 20)
print(f"N2 connected components + boxes of a 518x518 mask ({int(count.item())} components): {tc * 1e6:.0f} us on the device")

# ---- R1-R4 / N4: one CORAL validation step (features -> first-stage logits -> SparseRefiner -> thresholded full-size mask -> nine COD measures)
import types
from ucod_dpl_amd.engine.config import CfgNode
from ucod_dpl_amd.engine.runner import LocalRefineValidationLoop
from ucod_dpl_amd.models.uscod import baseline
from ucod_dpl_amd.models.UDLR import SparseRefiner
from ucod_dpl_amd.engine.utils.metrics import statistics
torch.manual_seed(0)
model = baseline(CfgNode(dict(dim=768, feature_size=68, ema_weight=0.99, dis_use_features=False))).to(dev).eval()
refiner = SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015))).eval().to(dev)
runner = types.SimpleNamespace(device=torch.device(dev, 0), model=model, refiner=refiner, world_size=1, rank=0, val_dataloader=[],
                               logger=types.SimpleNamespace(log_table=lambda *a: None, log=lambda *a: None))
for req_m in (False, True):
    cfg = CfgNode(dict(train_cfg=dict(dist_train=False), model_cfg=dict(window_length=6), dataset_cfg=dict(valset_cfg=dict(require_m_patches=req_m, DATASET="X"))))
    loop = LocalRefineValidationLoop(cfg, runner)
    g = torch.Generator().manual_seed(150)
    l = torch.randn(1, 768, 37, 37, generator=g).to(dev)
    full = torch.randn(1, 768, 54, 54, generator=g) * 0.7
    m = torch.stack([full[:, :, i * 18:i * 18 + 36, j * 18:j * 18 + 36] for i in range(2) for j in range(2)], dim=1).to(dev)
    h = torch.randn(1, 9, 768, 37, 37, generator=g).to(dev)
    label = (torch.rand(1, 1, 480, 640, generator=g) > 0.5).float().to(dev)
    batch = dict(pseudo_label=None, label_tensor=label, features=l, img_path=["x"], m_inputs=m, h_inputs=h, index=[0])
    st = statistics()
    t = sync_time(lambda: loop._process_validation_batch(batch, st), 10)
    print(f"N4 one CORAL validation step (require_m_patches={req_m}): first-stage decoder + SparseRefiner + 480x640 mask + nine COD measures: {t * 1e3:.2f} ms per image")

# ---- L1-L3 (BASELINE C4's validation side): Look-Twice on one image with the fallback centre box, real backbones
from ucod_dpl_amd.engine.runner import loop_look_twice as LT
rng = np.random.default_rng(0)
img_u8 = rng.integers(0, 256, (427, 640, 3), dtype=np.uint8)
for arch in ("dinov2_vitb14", "dinov2_vitl14"):
    bbx = backbone.random_init(arch, seed=0, image_size=518, device=dev, attn_variant=2)
    Dx = bbx.engine.D if hasattr(bbx.engine, "D") else (768 if arch.endswith("b14") else 1024)
    torch.manual_seed(5)
    mdl = baseline(CfgNode(dict(dim=Dx, feature_size=68, ema_weight=0.99, dis_use_features=False))).to(dev).eval()
    rn = types.SimpleNamespace(device=torch.device(dev, 0), model=mdl, world_size=1, rank=0, val_dataloader=[], logger=None)
    cfg = CfgNode(dict(train_cfg=dict(dist_train=False), model_cfg=dict(feature_size=68), val_cfg=dict(look_twice=True, look_twice_th=0.15, expand_type="dynamic"),
                       dataset_cfg=dict(valset_cfg=dict(image_size=(518, 518)))))
    lt = LT.ValLoop_Look_Twice(cfg, rn, feature_extractor=bbx)
    old = torch.zeros(1, 518, 518)
    for nbox in (1, 4):
        boxes = [list(LT.DEFAULT_BOX)] * nbox
        t = sync_time(lambda: lt.look_twice(img_u8, boxes, old), 10)
        print(f"L3 look_twice, {arch}, {nbox} box(es) of a 427x640 image (crop + Pillow-exact resize + backbone + decoder at 37x37 + bicubic paste): {t * 1e3:.2f} ms")
