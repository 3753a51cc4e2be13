#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):
    python tests/golden/make_golden.py
The reference has no tests or fixtures of its own, so these captured
input/output vectors are the parity pins for oracle/ and, through it, for the HIP path.
Import recipe = SURVEY.md Appendix B: stub the absent third-party modules (torchvision, timm,
cv2, prettytable, ntplib), skip models/__init__.py, and neutralise the hard-coded
``.to('cuda')``.  Only data (tensors in, tensors out) is written -- no reference source.
"""
import os
import sys
import types
import importlib.util
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from scipy import ndimage  # noqa: E402
import transformers  # noqa: E402,F401  (must be imported before torchvision is stubbed)
from transformers import Dinov2Config, Dinov2Model  # noqa: E402


def mod(n, **a):
    m = types.ModuleType(n)
    m.__dict__.update(a)
    sys.modules[n] = m
    return m


mod("models").__path__ = [REF + "/models"]


class DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()

    def forward(self, x):
        return x


mod("timm")
mod("timm.models")
mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x), trunc_normal_=nn.init.trunc_normal_)
mod("timm.models.registry", register_model=lambda f: f)


def connectedComponents(img, connectivity=8):
    """Stand-in for cv2.connectedComponents (OpenCV is not installed here): scipy's 8-connected partition, renumbered the way OpenCV numbers
    labels -- raster order of each component's first 2 x 2 block (cv2 labels 2 x 2 blocks in raster order, a block that touches nothing
    labelled yet takes the next provisional label, unions keep the smaller one, flattenL renumbers roots in increasing order)."""
    lab, n = ndimage.label(img > 0, structure=np.ones((3, 3)))
    if n:
        H, W = lab.shape
        ys, xs = np.nonzero(lab)
        key = (ys >> 1) * ((W + 1) >> 1) + (xs >> 1)
        first = np.full(n + 1, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(first, lab[ys, xs], key)
        order = np.argsort(first[1:], kind="stable")            # component ids (0-based) in OpenCV's order
        remap = np.zeros(n + 1, np.int32)
        remap[order + 1] = np.arange(1, n + 1, dtype=np.int32)
        lab = remap[lab]
    return n + 1, lab.astype(np.int32)


def boundingRect(m):
    ys, xs = np.nonzero(m)
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)


def connectedComponentsWithStats(img, connectivity=8):
    """(n, labels, stats[n,5] = LEFT, TOP, WIDTH, HEIGHT, AREA, centroids) like OpenCV, labels in raster order of first pixel."""
    n, lab = connectedComponents(img, connectivity)
    stats = np.zeros((n, 5), np.int32)
    cent = np.zeros((n, 2))
    for k in range(n):
        ys, xs = np.nonzero(lab == k)
        if len(ys):
            stats[k] = (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys))
            cent[k] = (xs.mean(), ys.mean())
    return n, lab, stats, cent


mod("cv2", connectedComponents=connectedComponents, boundingRect=boundingRect, connectedComponentsWithStats=connectedComponentsWithStats,
    CC_STAT_LEFT=0, CC_STAT_TOP=1, CC_STAT_WIDTH=2, CC_STAT_HEIGHT=3, CC_STAT_AREA=4)


class _T:
    def __init__(self, *a, **k):
        pass


mod("torchvision").transforms = mod("torchvision.transforms", Compose=_T, Resize=_T, ToTensor=_T, Normalize=_T, ToPILImage=_T)
mod("prettytable", PrettyTable=object)
mod("ntplib")
_to = torch.Tensor.to
torch.Tensor.to = lambda s, *a, **k: s if (a and a[0] == "cuda") else _to(s, *a, **k)

from engine.config.config import CfgNode  # noqa: E402
from models.uscod import baseline  # noqa: E402
from models.discriminator import Discriminator  # noqa: E402
import engine.runner.loop_UCOD_DPL as L  # noqa: E402


def npify(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def save(name, **d):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **npify(d))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def sd_flat(prefix, module):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


def model_cfg(dim, fs):
    return CfgNode(dict(dim=dim, feature_size=fs, ema_weight=0.99, dis_use_features=False))


# ----------------------------------------------------------------------------- G1: decoder fwd + grads
def g1():
    for tag, (B, C, H, W) in {"c384": (2, 384, 14, 14), "c768": (2, 768, 8, 8)}.items():
        torch.manual_seed(100 + C)
        m = baseline(model_cfg(C, H))
        x = torch.randn(B, C, H, W)
        r1, r2 = torch.randn(B, 1, H, W), torch.randn(B, 1, H, W)
        fg, bg, extra = m(x)
        teacher = m(x, ema=True)
        loss = (fg * r1).sum() + (bg * r2).sum() + 1000.0 * extra
        names = [n for n, _ in m.decoder.named_parameters()]
        grads = torch.autograd.grad(loss, list(m.decoder.parameters()), allow_unused=True)
        out = dict(x=x, r1=r1, r2=r2, fg=fg, bg=bg, extra=extra, teacher=teacher)
        out.update(sd_flat("sd.", m))
        for n, g in zip(names, grads):
            out["grad." + n] = torch.zeros(()) if g is None else g
        # fp64 run of the same module: pins the naive orthogonality loss for the Gram rewrite (G10)
        m64 = baseline(model_cfg(C, H)).double()
        m64.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
        _, _, extra64 = m64(x.double())
        out["extra_fp64"] = extra64
        save("g1_decoder_" + tag, **out)


# ----------------------------------------------------------------------------- G2: shipped checkpoints
def g2():
    from safetensors.torch import load_file
    for ver, C in (("dinov2", 768), ("dinov1", 384)):
        sd = load_file(f"{REF}/weights/UCOD_DPL_{ver}.safetensors")
        C = sd["decoder.decoupling.weight"].shape[1]
        m = baseline(model_cfg(C, 10))
        m.load_state_dict(sd, strict=True)
        b, c, h, w = torch.meshgrid(torch.arange(1.), torch.arange(float(C)), torch.arange(10.), torch.arange(10.), indexing="ij")
        x = torch.sin(0.37 * c + 1.3 * h + 0.7 * w) + 0.25 * torch.cos(0.011 * c * (h + 1) - 0.5 * w)
        fg, bg, extra = m(x)
        teacher = m(x, ema=True)
        # weights are NOT stored (they stay in the reference repo); keys/shapes are, for the strict-load test
        save("g2_shipped_" + ver, fg=fg, bg=bg, extra=extra, teacher=teacher,
             keys=np.array(sorted(sd.keys())), shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())]))


# ----------------------------------------------------------------------------- G3: discriminator
def g3():
    torch.manual_seed(3)
    d = Discriminator(model_cfg(768, 68))
    for p in d.parameters():
        p.requires_grad = True
    before = sd_flat("sd0.", d)
    mask = (torch.rand(4, 1, 68, 68) > 0.6).float()
    mask[:, :, 20:40, 10:50] = 1.0
    r = torch.randn(4, 1)
    prob = d(mask, None)
    loss = (prob * r).sum()
    names = [n for n, _ in d.named_parameters()]
    grads = torch.autograd.grad(loss, list(d.parameters()))
    out = dict(mask=mask, r=r, prob=prob)
    out.update(before)
    out.update(sd_flat("sd1.", d))
    for n, g in zip(names, grads):
        out["grad." + n] = g
    # second call on a different mask (running stats keep moving)
    mask2 = torch.zeros(4, 1, 68, 68)
    mask2[:, :, 5:30, 30:60] = 1.0
    out["mask2"] = mask2
    out["prob2"] = d(mask2, None).detach()
    out.update(sd_flat("sd2.", d))
    save("g3_discriminator", **out)


def g3b():
    """G3b: the REAL Discriminator with dis_use_features=True (models/discriminator.py:77-95): two forward calls (train-mode BatchNorm: running
    buffers keep moving) on seeded masks and feature maps; dim 32, feature_size 20 (-> 64 / 32 / 16 channels, Linear on 16 * 5 * 5)."""
    torch.manual_seed(33)
    d = Discriminator(CfgNode(dict(dim=32, feature_size=20, ema_weight=0.99, dis_use_features=True)))
    with torch.no_grad():                                   # non-trivial BatchNorm affine terms
        for n, p in d.named_parameters():
            if ".layers.1." in n:
                p.add_(0.2 * torch.randn_like(p))
    out = dict(sd_flat("sd0.", d))
    mask = (torch.rand(3, 1, 20, 20) > 0.6).float()
    feat = torch.randn(3, 32, 20, 20)
    with torch.no_grad():
        out["mask"], out["feature"], out["prob"] = mask, feat, d(mask, feat)
        out.update(sd_flat("sd1.", d))
        mask2 = torch.zeros(3, 1, 20, 20)
        mask2[:, :, 4:15, 2:12] = 1.0
        feat2 = torch.randn(3, 32, 20, 20) * 2 + 0.3
        out["mask2"], out["feature2"], out["prob2"] = mask2, feat2, d(mask2, feat2)
        out.update(sd_flat("sd2.", d))
    save("g3b_discriminator_features", **out)


# ----------------------------------------------------------------------------- fake runner for the loops
class NullLogger:
    def log(self, *a, **k):
        pass

    info = error = log_table = log


def make_loop(C, fs, seed, lr0=6e-4, dis_lr0=1e-3, dis_use_features=False):
    torch.manual_seed(seed)
    cfg = CfgNode(dict(
        model_cfg=dict(dim=C, feature_size=fs, ema_weight=0.99, dis_use_features=dis_use_features),
        train_cfg=dict(max_epoch=25, start_finetune=-5, lr0=lr0, dis_lr0=dis_lr0, step_lr_size=2, dis_step_lr_size=2,
                       step_lr_gamma=0.95, dis_step_lr_gamma=0.95, merge_alpha=0.5, dist_train=False, dis_epoch=1),
    ))
    model = baseline(cfg.model_cfg)
    disc = Discriminator(cfg.model_cfg)
    # the trained EMA teacher differs from the student: perturb it so the test can tell them apart
    with torch.no_grad():
        for p in model.decoder_ema.parameters():
            p.add_(0.05 * torch.randn_like(p))
    opt = torch.optim.AdamW(model.parameters(), lr=cfg.train_cfg.lr0)
    dis_opt = torch.optim.AdamW(disc.parameters(), lr=cfg.train_cfg.dis_lr0)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.train_cfg.step_lr_size, gamma=cfg.train_cfg.step_lr_gamma)
    dis_sched = torch.optim.lr_scheduler.StepLR(dis_opt, step_size=cfg.train_cfg.dis_step_lr_size, gamma=cfg.train_cfg.dis_step_lr_gamma)
    runner = SimpleNamespace(model=model, discriminator=disc, optimizer=opt, lr_scheduler=sched, dis_optimizer=dis_opt,
                             dis_lr_scheduler=dis_sched, logger=NullLogger(),
                             accelerator=SimpleNamespace(backward=lambda loss: loss.backward()))
    loop = L.TrainLoop.__new__(L.TrainLoop)
    loop.cfg = cfg
    loop._runner = runner
    loop._max_epoch = cfg.train_cfg.max_epoch
    loop._start_finetune = cfg.train_cfg.start_finetune
    loop._cur_epoch = 0
    loop.global_step = 0
    loop.finetune = False
    loop.criterion = nn.BCEWithLogitsLoss()
    loop.dis_loss = nn.BCELoss()
    loop.merge_alpha = cfg.train_cfg.merge_alpha
    loop.ema_alpha = cfg.model_cfg.ema_weight
    loop.progress_manager = SimpleNamespace(start_task=lambda *a: None, update_task=lambda *a: None, reset_task=lambda *a: None)
    return loop


def batch(B, C, seed):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, C, 14, 14, generator=g)
    pl = (torch.rand(B, 1, 16, 16, generator=g) > 0.7).float()
    return {"pseudo_label": pl, "label_tensor": torch.zeros(1), "features": feats, "img_path": ["x"]}


# ----------------------------------------------------------------------------- G4: APM merge
def g4():
    loop = make_loop(384, 28, seed=4)
    g = torch.Generator().manual_seed(44)
    pl = torch.rand(4, 1, 28, 28, generator=g)
    teacher = torch.randn(4, 1, 28, 28, generator=g) * 2
    student = torch.randn(4, 1, 28, 28, generator=g) * 2
    out = dict(pl=pl, teacher=teacher, student=student)
    out.update(sd_flat("disc0.", loop.runner.discriminator))
    for ep in (0, 10, 19, 20):
        loop._cur_epoch = ep
        merged, dl = loop.merge_pseudo_label(pl, teacher, student, None)
        out[f"merged_ep{ep}"] = merged
        out[f"dis_loss_ep{ep}"] = dl
        out.update(sd_flat(f"disc_after_ep{ep}.", loop.runner.discriminator))
    save("g4_apm_merge", **out)


# ----------------------------------------------------------------------------- G5: full training steps
def g5():
    loop = make_loop(384, 28, seed=5)
    out = {}
    out.update(sd_flat("model0.", loop.runner.model))
    out.update(sd_flat("disc0.", loop.runner.discriminator))
    for step in range(3):
        b = batch(4, 384, 500 + step)
        out[f"features{step}"] = b["features"]
        out[f"pl{step}"] = b["pseudo_label"]
        lr_before = loop.runner.optimizer.param_groups[0]["lr"]
        loss = loop._process_batch(b)
        loop.global_step += 1                             # run_epoch's own increment (loop_UCOD_DPL.py:143)
        out[f"loss{step}"] = loss.detach()
        out[f"lr_used{step}"] = np.float64(lr_before)
        for n, p in loop.runner.model.decoder.named_parameters():
            out[f"grad{step}.{n}"] = p.grad.clone() if p.grad is not None else torch.zeros(())
        out.update(sd_flat(f"model{step + 1}.", loop.runner.model))
        out.update(sd_flat(f"disc{step + 1}.", loop.runner.discriminator))
    save("g5_process_batch", **out)


# ----------------------------------------------------------------------------- G6: discriminator phase
def g6():
    loop = make_loop(384, 28, seed=6)
    d = loop.runner.discriminator
    out = {}
    out.update(sd_flat("model0.", loop.runner.model))
    out.update(sd_flat("disc0.", d))
    b = batch(4, 384, 600)
    out["features"] = b["features"]
    out["pl"] = b["pseudo_label"]
    loop.runner.train_dataloader = [b]
    losses = []
    loop.runner.logger = SimpleNamespace(log=lambda s, *a, **k: losses.append(s), info=lambda *a: None)
    for p in d.parameters():
        p.requires_grad = True
    loop.Discriminator_epoch()
    for n, p in d.named_parameters():
        out["grad." + n] = p.grad.clone()
    out["loss_str"] = np.array(losses)
    out.update(sd_flat("disc1.", d))
    save("g6_discriminator_step", **out)


def g6b():
    """G6b: the feature-branch discriminator (dis_use_features=True, models/discriminator.py:77-90) through the REAL loop: one
    Discriminator_epoch step (loop_UCOD_DPL.py:230-255: features resized 14 -> 12, both discriminator calls on them, BCE, backward, AdamW,
    StepLR) and, on the stepped module, one merge_pseudo_label (:257-272).  dim 16, feature_size 12 -> 48 / 24 / 12 channels, Linear 12 * 3 * 3."""
    loop = make_loop(16, 12, seed=66, dis_use_features=True)
    d = loop.runner.discriminator
    with torch.no_grad():                                   # non-trivial BatchNorm affine terms
        for n, p in d.named_parameters():
            if ".layers.1." in n:
                p.add_(0.2 * torch.randn_like(p))
    out = {}
    out.update(sd_flat("model0.", loop.runner.model))
    out.update(sd_flat("disc0.", d))
    b = batch(4, 16, 660)
    out["features"] = b["features"]
    out["pl"] = b["pseudo_label"]
    loop.runner.train_dataloader = [b]
    losses = []
    loop.runner.logger = SimpleNamespace(log=lambda s, *a, **k: losses.append(s), info=lambda *a: None)
    for p in d.parameters():
        p.requires_grad = True
    loop.Discriminator_epoch()
    for n, p in d.named_parameters():
        out["grad." + n] = p.grad.clone()
    out["loss_str"] = np.array(losses)
    out.update(sd_flat("disc1.", d))
    # APM merge with the feature branch (what _process_batch calls): 12 x 12 logits and features
    g = torch.Generator().manual_seed(661)
    pl = torch.rand(4, 1, 12, 12, generator=g)
    teacher, student = torch.randn(4, 1, 12, 12, generator=g) * 2, torch.randn(4, 1, 12, 12, generator=g) * 2
    feats = torch.randn(4, 16, 12, 12, generator=g)
    loop._cur_epoch = 10
    with torch.no_grad():
        merged, dl = loop.merge_pseudo_label(pl, teacher, student, feats)
    out.update(dict(apm_pl=pl, apm_teacher=teacher, apm_student=student, apm_features=feats, apm_merged=merged, apm_dis_loss=dl))
    out.update(sd_flat("disc2.", d))
    save("g6b_discriminator_features_step", **out)


# ----------------------------------------------------------------------------- G7: Look-Twice integer table
def g7():
    rng = np.random.default_rng(7)
    V = L.ValLoop_Look_Twice
    masks, boxes_dyn, boxes_const = [], [], []
    H = W = 64

    def blob(m, cy, cx, ry, rx):
        yy, xx = np.mgrid[0:H, 0:W]
        m[((yy - cy) / max(ry, 1)) ** 2 + ((xx - cx) / max(rx, 1)) ** 2 <= 1.0] = 1

    cases = []
    for i in range(240):
        m = np.zeros((H, W), np.uint8)
        kind = i % 6
        if kind == 0 and i < 12:
            pass                                         # empty
        elif kind == 1:
            blob(m, rng.integers(10, 54), rng.integers(10, 54), rng.integers(12, 30), rng.integers(12, 30))   # one large
        elif kind == 2:
            for _ in range(rng.integers(2, 7)):
                blob(m, rng.integers(0, H), rng.integers(0, W), rng.integers(2, 7), rng.integers(2, 7))       # many small
        elif kind == 3:
            blob(m, rng.integers(0, 4), rng.integers(0, W), rng.integers(3, 9), rng.integers(3, 9))           # edge-touching
            blob(m, rng.integers(H - 4, H), rng.integers(0, W), rng.integers(3, 9), rng.integers(3, 9))
        elif kind == 4:
            blob(m, rng.integers(20, 60), rng.integers(5, 60), rng.integers(3, 6), rng.integers(3, 10))       # low & small -> br/fr big
        else:
            m[rng.integers(0, H, 40), rng.integers(0, W, 40)] = 1                                             # speckle
            blob(m, rng.integers(8, 56), rng.integers(8, 56), rng.integers(4, 8), rng.integers(4, 8))
        cases.append(m)
    # Label-ORDER cases (appended, so the 240 masks above keep their indices): equal rectangles -> equal box areas -> process_preds' stable
    # sort keeps the labelling's order (:382).  Placed so that OpenCV's order (first 2 x 2 block in raster order) differs from the order of
    # first pixels: the rectangle on the left starts on the ODD row of a block row, the one on the right on its even row.
    for (ya, xa, yb, xb, hh, ww) in ((1, 2, 0, 40, 8, 8), (11, 4, 10, 30, 7, 9), (21, 0, 20, 50, 10, 6), (3, 20, 2, 44, 8, 8), (31, 6, 30, 36, 9, 9),
                                     (41, 10, 40, 48, 8, 10), (1, 30, 0, 50, 8, 8), (51, 2, 50, 34, 8, 8)):
        mm = np.zeros((H, W), np.uint8)
        mm[ya:ya + hh, xa:xa + ww] = 1
        mm[yb:yb + hh, xb:xb + ww] = 1
        cases.append(mm)
        mm2 = mm.copy()
        mm2[(ya + 24) % 50:(ya + 24) % 50 + hh, 22:22 + ww] = 1       # a third equal rectangle elsewhere
        cases.append(mm2)
    fake = SimpleNamespace(img_size=(H, W), cfg=SimpleNamespace(val_cfg=SimpleNamespace(look_twice_th=0.15, expand_type="dynamic")))
    fake.expand_bbox = lambda *a, **k: V.expand_bbox(fake, *a, **k)
    recs = []
    for m in cases:
        # logits whose upsample-to-(H,W) is the identity: feed at full resolution
        logits = torch.from_numpy(m.astype(np.float32) * 8 - 4).view(1, 1, H, W)
        row = {"mask": m}
        for et in ("dynamic", "const"):
            fake.cfg.val_cfg.expand_type = et
            try:
                _, bx = V.process_preds(fake, logits, None)
                row[et] = "none" if bx is None else ";".join(",".join(str(v) for v in b) for b in bx)
            except ValueError:
                row[et] = "ValueError"
            except ZeroDivisionError:
                row[et] = "ZeroDivisionError"
        recs.append(row)
    rb = []
    for _ in range(64):
        b = [int(v) for v in rng.integers(0, 400, 4)]
        ow, oh, nw, nh = (int(v) for v in rng.integers(100, 2000, 4))
        rb.append(b + [ow, oh, nw, nh] + V.resize_bbox(None, b, ow, oh, nw, nh))
    save("g7_look_twice_int", masks=np.stack([r["mask"] for r in recs]), dynamic=np.array([r["dynamic"] for r in recs]),
         const=np.array([r["const"] for r in recs]), resize_bbox=np.array(rb, np.int64))


# ----------------------------------------------------------------------------- G8: ViT backbones
def g8():
    # DINOv2 (HF): D=128, 2 heads x hd 64, 3 layers; native grid (no pos interpolation) and interpolated grid
    for tag, (img, pre) in {"native": (70, 70), "interp": (70, 56)}.items():
        torch.manual_seed(8)
        cfg = Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=pre, patch_size=14,
                           mlp_ratio=4, layerscale_value=1.0)
        m = Dinov2Model(cfg).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():              # non-trivial LN / layerscale / bias values
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
                if "position_embeddings" in n or "cls_token" in n:
                    p.mul_(0.05)
        keys = {}
        m.encoder.layer[-1].attention.attention.key.register_forward_hook(lambda mod_, i, o: keys.__setitem__("k", o.detach()))
        x = torch.randn(2, 3, img, img)
        with torch.no_grad():
            out = m(x)
        k = keys["k"]
        B, Ntok, C = k.shape
        g = int((Ntok - 1) ** 0.5)
        kmap = k[:, 1:, :].reshape(B, g, g, C).permute(0, 3, 1, 2)
        d = dict(x=x, last_hidden_state=out.last_hidden_state, key=kmap)
        d.update({"sd." + n: v for n, v in m.state_dict().items()})
        save("g8_dinov2_" + tag, **d)
    # DINOv1 (in-repo dino.py)
    spec = importlib.util.spec_from_file_location("ref_dino", REF + "/models/backbones/dino.py")
    dino = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dino)
    from functools import partial
    for tag, (img, pre) in {"native": (32, 32), "interp": (48, 32)}.items():
        torch.manual_seed(81)
        m = dino.VisionTransformer(img_size=[pre], patch_size=8, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4,
                                   qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        feat = {}
        m.blocks[-1].attn.qkv.register_forward_hook(lambda mod_, i, o: feat.__setitem__("qkv", o.detach()))
        x = torch.randn(2, 3, img, img)
        with torch.no_grad():
            tok = m.prepare_tokens(x)
            for blk in m.blocks:
                tok = blk(tok)
            last = m.norm(tok)
        qkv = feat["qkv"]
        B, Ntok, _ = qkv.shape
        k = qkv.reshape(B, Ntok, 3, 2, 64).permute(2, 0, 3, 1, 4)[1].transpose(1, 2).reshape(B, Ntok, 128)
        g = img // 8
        kmap = k[:, 1:].transpose(1, 2).reshape(B, 128, g, g)                  # ViTFeat 'k' (dino.py:308-320)
        d = dict(x=x, last_hidden_state=last, key=kmap)
        d.update({"sd." + n: v for n, v in m.state_dict().items()})
        save("g8_dinov1_" + tag, **d)


# ----------------------------------------------------------------------------- G9: CORAL SparseRefiner (eval forward)
def g9():
    from models.UDLR import SparseRefiner
    sys.path.insert(0, OUT)
    import refiner_init as RI
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015, dim=768)))).eval()
    out = {"chk." + k: v for k, v in RI.checksums(m).items()}
    for tag, partial in (("full", False), ("partial", True)):
        l, h, preds = RI.make_inputs(partial)
        with torch.no_grad():
            outputs, ex, opt = m(l, h, preds)
        out[tag + ".outputs"] = outputs
        for k in ("mask", "entropy", "h_preds", "window_preds", "GE_w", "coords_list"):
            out[f"{tag}.{k}"] = opt[k].float() if opt[k].dtype == torch.bool else opt[k]
        out[tag + ".ex_loss"] = np.float64(ex)
    save("g9_refiner", **out)


def g9b():
    """G9b: the REAL SparseRefiner in .train() mode with h_targets (models/UDLR.py:52-86): the training-mode forward (identical arithmetic:
    every dropout is 0) and cal_ex_loss's IoU-weighted window loss, for soft targets in [0,1] and for logit targets (binary_iou's
    "max > 1" branch), full and partial window selections."""
    from models.UDLR import SparseRefiner
    sys.path.insert(0, OUT)
    import refiner_init as RI
    torch.manual_seed(RI.SEED)
    m = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015, dim=768)))).train()
    out = {}
    for tag, partial in (("full", False), ("partial", True)):
        l, h, preds = RI.make_inputs(partial)
        for kind in ("prob", "logit"):
            ht = RI.make_h_targets(kind)
            with torch.no_grad():
                outputs, ex, opt = m(l, h, preds, ht)
            k = f"{tag}.{kind}."
            out[k + "ex_loss"] = torch.as_tensor(ex).double()
            out[k + "window_targets"] = opt["window_targets"]
            out[k + "outputs"] = outputs
            out[k + "window_preds"] = opt["window_preds"]
    save("g9b_refiner_train", **out)


# ----------------------------------------------------------------------------- G12: backbone backward (LoRA mode, row B9)
class _LoRALinear(nn.Module):
    """What peft's LoRA wrapper computes for the LoraConfig of models/modules/full_model.py:47-72 (r, lora_alpha, bias='none';
    peft is not installed here, so its published forward is restated; dropout left out = eval / p=0):
    base(x) + lora_B(lora_A(x)) * lora_alpha / r, lora_A kaiming_uniform(a=sqrt(5)), lora_B zeros."""

    def __init__(self, base, r, alpha):
        super().__init__()
        self.base = base
        self.lora_A = nn.Linear(base.in_features, r, bias=False)
        self.lora_B = nn.Linear(r, base.out_features, bias=False)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=5 ** 0.5)
        nn.init.zeros_(self.lora_B.weight)
        self.scaling = alpha / r

    def forward(self, x):
        return self.base(x) + self.lora_B(self.lora_A(x)) * self.scaling


def g12():
    import torch.nn.functional as F
    # (a) HF Dinov2 + LoRA on query/key/value, the hook of full_model.py:95-106 (key -> drop CLS -> NCHW -> bilinear), autograd
    torch.manual_seed(12)
    cfg = Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=70, patch_size=14, mlp_ratio=4,
                       layerscale_value=1.0)
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            if "position_embeddings" in n or "cls_token" in n:
                p.mul_(0.05)
    for p in m.parameters():
        p.requires_grad_(False)
    r, alpha = 2, 4
    for layer in m.encoder.layer:
        att = layer.attention.attention
        for name in ("query", "key", "value"):
            w = _LoRALinear(getattr(att, name), r, alpha)
            with torch.no_grad():
                w.lora_B.weight.copy_(0.05 * torch.randn_like(w.lora_B.weight))     # non-zero B: every LoRA gradient is exercised
            setattr(att, name, w)
    keys = {}
    m.encoder.layer[-1].attention.attention.key.register_forward_hook(lambda mod_, i, o: keys.__setitem__("k", o))
    x = torch.randn(2, 3, 70, 70)
    m(x)
    k = keys["k"]
    B, Ntok, C = k.shape
    g = int((Ntok - 1) ** 0.5)
    kmap = k[:, 1:, :].reshape(B, g, g, C).permute(0, 3, 1, 2)
    kmap.retain_grad()
    up = F.interpolate(kmap, size=(8, 8), mode="bilinear")
    R = torch.randn_like(up)
    (up * R).sum().backward()
    d = dict(x=x, R=R, key=kmap.detach(), dkey=kmap.grad, lora_scale=np.float64(alpha / r))
    for n, v in m.state_dict().items():
        n = n.replace(".base.", ".")
        d["sd." + n] = v
    for n, p in m.named_parameters():
        if "lora_" in n:
            d["grad." + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    save("g12_lora_backbone", **d)
    # (b) the REAL reference decoder: gradient w.r.t. its input features (what flows back into the key hook)
    torch.manual_seed(1200)
    C_, H_ = 768, 8
    dec = baseline(model_cfg(C_, H_))
    feat = torch.randn(2, C_, H_, H_, requires_grad=True)
    r1, r2 = torch.randn(2, 1, H_, H_), torch.randn(2, 1, H_, H_)
    fg, bg, extra = dec(feat)
    ((fg * r1).sum() + (bg * r2).sum() + 1000.0 * extra).backward()
    out = dict(x=feat.detach(), r1=r1, r2=r2, dx=feat.grad)
    out.update(sd_flat("sd.", dec))
    save("g12_decoder_dx", **out)


# ----------------------------------------------------------------------------- G13: feature-cache on-disk format (row N1)
def g13():
    """A three-item features cache written by the reference's OWN MultiCacheManager / MetaListPickleIO
    (data/datasets/cache_manager.py, engine/utils/fileio/backend/ioctl/pickleio.py) -> tests/golden/cache_ref/."""
    import shutil
    from data.datasets.cache_manager import MultiCacheManager
    root = os.path.join(OUT, "cache_ref")
    shutil.rmtree(root, ignore_errors=True)
    log = SimpleNamespace(log=lambda *a, **k: None)
    m = MultiCacheManager(root, "dinov2", "train", "COD10K", logger=log)
    g = torch.Generator().manual_seed(13)
    feats = [torch.randn(4, 3, 3, generator=g) for _ in range(3)]
    m.get_features_cache().dump_list(feats)
    m.get_pseudo_label_cache().dump_list([(f[:1] > 0).float() for f in feats])
    save("g13_cache_items", **{f"f{i}": f for i, f in enumerate(feats)})


# ----------------------------------------------------------------------------- G14: pseudo-label generator (row N3)
def g14():
    """data/utils/found_bkg_mask.py::compute_img_bkg_seg and generate_pseudo_label.py::refine_post_process -- the REAL reference
    functions -- on the last-layer attentions / key hook of a seeded HF Dinov2Model (eager attention, output_attentions)."""
    from data.utils.found_bkg_mask import compute_img_bkg_seg
    mod("tqdm", tqdm=lambda x, **k: x)
    spec = importlib.util.spec_from_file_location("ref_gpl", REF + "/generate_pseudo_label.py")
    gpl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gpl)
    torch.manual_seed(14)
    cfg = Dinov2Config(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=112, patch_size=14, mlp_ratio=4,
                       layerscale_value=1.0)
    cfg._attn_implementation = "eager"
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            if "position_embeddings" in n or "cls_token" in n:
                p.mul_(0.05)
    keys = {}
    m.encoder.layer[-1].attention.attention.key.register_forward_hook(lambda mod_, i, o: keys.__setitem__("k", o.detach()))
    x = torch.randn(3, 3, 112, 112)
    x[1] = x[1] * 0.2 + torch.linspace(-2, 2, 112).view(1, 1, 112)          # a smoother image: different sparsity pattern
    with torch.no_grad():
        out = m(x, output_attentions=True)
    attn = out.attentions[-1]
    key = keys["k"]
    nh = attn.shape[1]
    d = dict(x=x, attn_cls=attn[:, :, 0, :].clone(), key=key)
    d.update({"sd." + n: v for n, v in m.state_dict().items()})
    for th in (0.6, 0.3):
        for aw in (True, False):
            mask, sim = compute_img_bkg_seg(attentions=attn, feats=key, featmap_dims=(8, 8), th_bkg=th, dim=key.shape[-1] // nh, apply_weights=aw)
            tag = f"th{int(th * 10)}_w{int(aw)}"
            d["mask." + tag], d["sim." + tag] = mask, sim
    # refine_post_process on binary masks with small islands / holes of every kind
    g = torch.Generator().manual_seed(15)
    masks = (torch.rand(24, 1, 16, 16, generator=g) > 0.82).float()
    masks[8:16] = 1 - masks[8:16]
    masks[16:] = (torch.rand(8, 1, 16, 16, generator=g) > 0.5).float()
    d["pp_in"] = masks
    d["pp_out"] = torch.stack([gpl.refine_post_process(mk.clone()) for mk in masks])
    d["pp_out_a9"] = torch.stack([gpl.refine_post_process(mk.clone(), area_threshold=9) for mk in masks])
    save("g14_pseudo_label", **d)


# ----------------------------------------------------------------------------- G15: CORAL validation loop pieces (row N4)
def g15():
    """engine/runner/loop_CORAL.py::LocalRefineValidationLoop -- the REAL class (constructed without its progress-bar __init__) on
    the real `baseline` and `SparseRefiner`: feature preparation (both require_m_patches settings), crop decision, refiner call,
    centre padding and prediction post-processing.  Inputs come from tests/golden/refiner_init.coral_inputs (seeded)."""
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches", "torchvision.transforms.functional"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    sys.modules["torchvision"].transforms.functional = sys.modules["torchvision.transforms.functional"]
    import engine.runner.loop_CORAL as LC
    from models.UDLR import SparseRefiner
    sys.path.insert(0, OUT)
    import refiner_init as RI
    torch.manual_seed(RI.SEED)
    refiner = RI.perturb_(SparseRefiner.from_config(CfgNode(dict(window_size=3, threshold=0.0015, dim=768)))).eval()
    torch.manual_seed(151)
    model = baseline(model_cfg(768, 68)).eval()
    out = {"dec." + k: v for k, v in model.state_dict().items()}
    for req_m in (False, True):
        loop = object.__new__(LC.LocalRefineValidationLoop)
        loop.cfg = CfgNode(dict(model_cfg=dict(window_length=6), dataset_cfg=dict(valset_cfg=dict(require_m_patches=req_m, DATASET="X"))))
        loop._runner = SimpleNamespace(model=model, refiner=refiner)
        loop.window_length = 6
        l, m, h = RI.coral_inputs()
        with torch.no_grad():
            fd = loop._prepare_validation_features(l, m, h)
            crop = loop._should_crop_center(fd["preds"])
            outputs, _, opt = refiner(fd["l_features"], fd["h_features"], fd["preds"])
            padded = loop._center_pad(outputs)
            up = loop.process_preds(outputs, (50, 70))
            up_pad = loop.process_preds(padded, (50, 70))
        t = f"m{int(req_m)}."
        out.update({t + "l_features": fd["l_features"], t + "h_features": fd["h_features"], t + "preds": fd["preds"], t + "crop": np.int64(bool(crop)),
                    t + "outputs": outputs, t + "padded": padded, t + "up": up, t + "up_pad": up_pad})
    # _should_crop_center on both sides of its 0.001 threshold, and process_preds on already-probabilities
    loop = object.__new__(LC.LocalRefineValidationLoop)
    z = torch.full((1, 1, 40, 40), -1.0)
    z[0, 0, 0, 0] = 1.0
    out["crop_sparse"] = np.int64(bool(loop._should_crop_center(z)))
    z[0, 0, 0, :2] = 1.0
    out["crop_dense"] = np.int64(bool(loop._should_crop_center(z)))
    pr = torch.rand(1, 1, 9, 9, generator=torch.Generator().manual_seed(3))
    out["probs_in"] = pr
    out["probs_up"] = loop.process_preds(pr, (20, 31))
    save("g15_coral_loop", **out)


# ----------------------------------------------------------------------------- G16: COD metrics (row N4, `statistics`)
def g16():
    """engine/utils/metrics/metric.py::statistics -- the REAL class (cv2, which it imports for its file-based helper only, is stubbed)
    on seeded prediction / ground-truth pairs: soft and binary predictions, blobs, empty and full ground truths, constant
    predictions, odd sizes.  Stored per case: every per-image quantity the measures append, and the final get_result() of the set."""
    if "cv2" not in sys.modules:
        mod("cv2")
    import engine.utils.metrics.metric as M
    g = torch.Generator().manual_seed(16)

    def blob(h, w, n):
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        m = torch.zeros(h, w)
        for _ in range(n):
            cy, cx = torch.rand(1, generator=g).item() * h, torch.rand(1, generator=g).item() * w
            ry, rx = 2 + torch.rand(1, generator=g).item() * h / 3, 2 + torch.rand(1, generator=g).item() * w / 3
            m = torch.maximum(m, (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1).float())
        return m

    cases = []
    for (h, w) in ((23, 31), (40, 40), (17, 52), (64, 48)):
        gt = blob(h, w, 2)
        soft = (gt * 0.6 + torch.rand(h, w, generator=g) * 0.5).clamp(0, 1)                    # soft prediction correlated with gt
        cases.append((gt, soft))
        cases.append((gt, (soft > 0.55).float()))                                               # binary prediction (what the loops pass)
        cases.append((gt * 255.0, torch.rand(h, w, generator=g) * 3.0 - 1.0))                   # 0/255 gt, unnormalised prediction
    cases.append((torch.zeros(20, 20), torch.rand(20, 20, generator=g)))                         # empty ground truth
    cases.append((torch.ones(20, 20), torch.rand(20, 20, generator=g)))                          # full ground truth
    cases.append((blob(20, 26, 1), torch.full((20, 26), 0.7)))                                   # constant prediction (astype(int) branch)
    cases.append((blob(20, 26, 1), torch.ones(20, 26)))
    cases.append(((torch.rand(30, 30, generator=g) > 0.9).float(), torch.rand(30, 30, generator=g)))   # scattered gt: many EDT ties
    cases.append(((torch.rand(15, 33, generator=g) > 0.5).float(), (torch.rand(15, 33, generator=g) > 0.5).float()))
    out = {"n": np.int64(len(cases))}
    st = M.statistics()
    for i, (gt, pred) in enumerate(cases):
        out[f"gt{i}"], out[f"pred{i}"] = gt, pred
        st.step(gt.unsqueeze(0).unsqueeze(0), pred.unsqueeze(0).unsqueeze(0))
        out[f"mae{i}"] = np.float64(st.MAE.maes[-1])
        out[f"acc{i}"] = np.float64(st.ACC.accs[-1])
        out[f"iou{i}"] = np.float64(st.MIOU.ious[-1])
        out[f"sm{i}"] = np.float64(st.SM.sms[-1])
        out[f"wfm{i}"] = np.float64(st.WFM.weighted_fms[-1])
        out[f"adp_em{i}"] = np.float64(st.EM.adaptive_ems[-1])
        out[f"em_curve{i}"] = np.asarray(st.EM.changeable_ems[-1], np.float64)
        out[f"adp_fm{i}"] = np.float64(st.FM.adaptive_fms[-1])
        out[f"fm_curve{i}"] = np.asarray(st.FM.changeable_fms[-1], np.float64)
        out[f"p_curve{i}"] = np.asarray(st.FM.precisions[-1], np.float64)
        out[f"r_curve{i}"] = np.asarray(st.FM.recalls[-1], np.float64)
    for k, v in st.get_result().items():
        out["final." + k] = np.float64(v)
    save("g16_cod_metrics", **out)


# ----------------------------------------------------------------------------- G11: the reference's own look_twice composition
class _PILResize:
    """torchvision.transforms.Resize((h, w)) on a PIL image: Image.resize((w, h), BILINEAR) (torchvision/transforms/functional_pil.py)."""

    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        from PIL import Image
        return img.resize((self.size[1], self.size[0]), Image.BILINEAR)


class _PILToTensor:
    """torchvision.transforms.ToTensor on a PIL image: uint8 HWC -> float CHW / 255."""

    def __call__(self, img):
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return t.clone().sub_(self.mean).div_(self.std)


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _ToPILImage:
    """torchvision.transforms.ToPILImage on a float tensor [H,W] / [1,H,W]: mul(255).byte() -> mode 'L'."""

    def __call__(self, t):
        from PIL import Image
        if t.dim() == 2:
            t = t.unsqueeze(0)
        a = t.mul(255).byte().numpy().transpose(1, 2, 0)
        return Image.fromarray(a[:, :, 0], mode="L")


G11_SEED = [1111]


def g11():
    """ValLoop_Look_Twice.look_twice (engine/runner/loop_UCOD_DPL.py:326-352) run as the reference wrote it: the real method, the real
    ``backbone.forward`` (data/utils/feature_extractor.py:49-59) over the seeded HF Dinov2Model of G8 (weights already in
    g8_dinov2_native.npz), the real ``baseline`` decoder; torchvision's four transforms replaced by their Pillow definitions."""
    import tempfile
    from PIL import Image
    from data.utils.feature_extractor import backbone as RefBackbone
    V = L.ValLoop_Look_Twice
    g8 = np.load(os.path.join(OUT, "g8_dinov2_native.npz"))
    cfg = Dinov2Config(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, image_size=70, patch_size=14, mlp_ratio=4, layerscale_value=1.0)
    vit = Dinov2Model(cfg).eval()
    vit.load_state_dict({k[3:]: torch.from_numpy(g8[k]) for k in g8.files if k.startswith("sd.")}, strict=True)
    fe = RefBackbone.__new__(RefBackbone)                  # __init__ downloads a checkpoint and calls .cuda(): build the same object by hand
    nn.Module.__init__(fe)
    fe.config = SimpleNamespace(backbone="facebook/dinov2-base")
    fe.feature_extractor, fe.key = vit, None
    vit.encoder.layer[-1].attention.attention.key.register_forward_hook(fe.hook_fn_key)

    torch.manual_seed(G11_SEED[0])
    model = baseline(model_cfg(128, 5)).eval()
    with torch.no_grad():                                   # decisive logits: no pixel of the 5x5 predictions sits near the threshold
        model.decoder.conv_out_fg.weight.mul_(40.0)
    ih, iw = 70, 70
    rng = np.random.default_rng(11)
    H, W = 427, 640
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 / W), (yy * 255 / H), ((xx * 3 + yy * 2) % 256)], -1).astype(np.float32) + rng.normal(0, 25, (H, W, 3))
    for _ in range(8):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(15, 90)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] += rng.integers(-140, 140, 3)
    img = np.clip(img, 0, 255).astype(np.uint8)
    old = torch.zeros(1, ih, iw)
    old[:, 20:41, 8:30] = 1.0
    bboxes = [[6, 10, 40, 33], [38, 30, 27, 36], [-4, 44, 30, 22], [50, 2, 26, 20]]   # x,y,w,h in img_size pixels; one leaves the image
    rec = dict(crops=[], logits=[])

    class FE:
        def __call__(self, x):
            rec["crops"].append(x.clone())
            return fe(x)

    class M:
        def __call__(self, f):
            out = model(f)
            rec["logits"].append(out[0].detach().clone())
            return out

    fake = SimpleNamespace(img_size=(ih, iw), feature_extractor=FE(), runner=SimpleNamespace(model=M()), to_PIL=_ToPILImage(), to_tensor=_PILToTensor(),
                           transform_image=_Compose([_PILResize((ih, iw)), _PILToTensor(), _Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])]))
    fake.resize_bbox = lambda *a: V.resize_bbox(fake, *a)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "img.png")
        Image.fromarray(img).save(path)
        V.look_twice(fake, path, bboxes, old.clone())      # probe run: centre the logits so that each crop predicts both classes
        with torch.no_grad():
            v = torch.cat(rec["logits"]).flatten().sort().values
            lo, hi = int(0.35 * v.numel()), int(0.65 * v.numel())
            k = lo + int((v[lo + 1:hi + 1] - v[lo:hi]).argmax())                 # widest gap in the middle of the distribution
            model.decoder.conv_out_fg.bias.sub_((v[k] + v[k + 1]) / 2)
        rec["crops"].clear()
        rec["logits"].clear()
        new_mask = V.look_twice(fake, path, bboxes, old.clone())
    logits = torch.cat(rec["logits"])
    assert sum(0 < (l > 0).sum() < l.numel() for l in logits) >= 3, [(l > 0).sum().item() for l in logits]
    margin = logits.abs().min().item()
    if margin < 0.3:                                        # the bf16 device path must land on the same side: try the next decoder seed
        G11_SEED[0] += 1
        return g11()
    d = dict(image=img, old_mask=old, bboxes=np.asarray(bboxes, np.int64), crops=torch.cat(rec["crops"]), logits=logits, new_mask=new_mask,
             logit_margin=np.float64(margin), decoder_seed=np.int64(G11_SEED[0]))
    d.update({"sd." + k: v for k, v in model.state_dict().items()})
    save("g11_look_twice", **d)


# ----------------------------------------------------------------------------- G17: the reference's config trees
def g17():
    import json

    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return {"__tuple__" if isinstance(x, tuple) else "__list__": [plain(v) for v in x]}
        return x

    trees = {}
    for rel in ("uscod/UCOD-DPL_dinov1.py", "uscod/UCOD-DPL_dinov2.py", "uscod/CORAL_dinov1.py", "uscod/CORAL_dinov2.py",
                "__base__/newbase.py", "__base__/accelerate.py", "dataset/cod4040.py"):
        trees[rel] = plain(dict(CfgNode(CfgNode.load_with_base(os.path.join(REF, "configs", rel)))))
    path = os.path.join(OUT, "g17_config_trees.json")
    with open(path, "w") as f:
        json.dump(trees, f, indent=1, sort_keys=True)
    print("wrote", path)


# ----------------------------------------------------------------------------- G18: the epoch-level schedule, from the reference's own TrainLoop.run()
def g18():
    """The REAL ``TrainLoop.run()`` (engine/runner/loop_UCOD_DPL.py:94-118) over 6 epochs of a 3-batch in-memory loader with
    start_finetune = -2 (finetune switch at epoch 4: ``runner.start_finetune()`` REBUILDS both optimisers and schedulers, runner.py:378-379 -> :276-311;
    ``global_step`` back to 0, :101-103), dis_intertrain = 2 / dis_epoch = 1 (discriminator phase before epochs 0 and 2 with the requires_grad flips,
    :193-227; none once finetune is on), ``loss -= dis_loss`` only before finetune (:167-169), StepLR stepped per batch, EMA ramp on the doubled
    ``global_step``.  Validation and saving are off.  ``start_finetune`` is the reference's own ``StandardRunner.start_finetune`` / ``_build_optimizer``
    bound to the fake runner.  Recorded after every discriminator phase and every epoch: decoder, EMA decoder, discriminator (parameters + BatchNorm
    buffers), both learning rates, ``global_step``, ``finetune``; and every batch's loss."""
    C_, fs, B, nb = 128, 12, 4, 3
    torch.manual_seed(18)
    cfg = CfgNode(dict(
        model_cfg=dict(dim=C_, feature_size=fs, ema_weight=0.99, dis_use_features=False),
        train_cfg=dict(max_epoch=6, start_epoch=0, start_finetune=-2, lr0=6e-4, dis_lr0=1e-3, step_lr_size=2, dis_step_lr_size=2, step_lr_gamma=0.95,
                       dis_step_lr_gamma=0.95, merge_alpha=0.5, merge_method="dis", dist_train=False, dis_epoch=1, dis_intertrain=2,
                       save_cfg=dict(save_mode="model", save_interval=5, start_save=1000)),
        val_cfg=dict(enable_val=False, val_interval=5, start_val=1000),
        log_cfg=dict(log_interval=50, log_path="/tmp/ucod_g18", multi_rank=[0]),
    ))
    model, disc = baseline(cfg.model_cfg), Discriminator(cfg.model_cfg)
    with torch.no_grad():
        for p in model.decoder_ema.parameters():
            p.add_(0.05 * torch.randn_like(p))
    g = torch.Generator().manual_seed(1800)
    loader = []
    for _ in range(nb):
        feats = torch.randn(B, C_, 14, 14, generator=g)
        pl = (torch.rand(B, 1, 16, 16, generator=g) > 0.7).float()
        loader.append({"pseudo_label": pl, "label_tensor": torch.zeros(1), "features": feats, "img_path": ["x"]})
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.patches", "torchvision.transforms.functional"):      # (what runner.py's import of loop_CORAL pulls in: as in g15)
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    sys.modules["torchvision"].transforms.functional = sys.modules["torchvision.transforms.functional"]
    import engine.runner.runner as R                                   # the reference's runner module (accelerate is installed)
    losses, events = [], []

    class Log(NullLogger):
        def log(self, s, *a, **k):
            if isinstance(s, str) and s.startswith("iter") and ":loss:" in s:
                losses.append(s)

    runner = SimpleNamespace(config=cfg, model=model, discriminator=disc, logger=Log(), train_dataloader=loader, val_dataloader=[],
                             accelerator=SimpleNamespace(backward=lambda loss: loss.backward()))
    runner.logger.info = runner.logger.error = lambda *a, **k: None
    runner._build_optimizer = lambda: R.StandardRunner._build_optimizer(runner)
    runner._build_optimizer()                                           # runner.py:276-311
    runner.start_finetune = lambda: R.StandardRunner.start_finetune(runner)   # runner.py:378-379
    loop = L.TrainLoop.__new__(L.TrainLoop)
    loop.cfg, loop._runner = cfg, runner
    loop._dist_train = False
    loop._mode = "train"
    loop._start_epoch, loop._max_epoch = cfg.train_cfg.start_epoch, cfg.train_cfg.max_epoch
    loop.global_step, loop._cur_epoch = 0, 0
    loop._start_finetune, loop.finetune = cfg.train_cfg.start_finetune, False
    loop.criterion, loop.dis_loss = nn.BCEWithLogitsLoss(), nn.BCELoss()
    loop.merge_alpha, loop.ema_alpha = cfg.train_cfg.merge_alpha, cfg.model_cfg.ema_weight
    L.TrainLoop._setup_validation_config(loop)
    L.TrainLoop._setup_logging_config(loop)
    loop.best_mae, loop.best_result = 1000.0, None

    class PM:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def start_task(self, *a):
            pass

        update_task = reset_task = start_task

    loop.progress_manager = PM()
    out = {}
    out.update(sd_flat("model0.", model))
    out.update(sd_flat("disc0.", disc))
    for i, b in enumerate(loader):
        out[f"features{i}"], out[f"pl{i}"] = b["features"], b["pseudo_label"]

    def snap(tag):
        events.append(tag)
        out.update(sd_flat(f"{tag}.model.", model))
        out.update(sd_flat(f"{tag}.disc.", disc))
        out[f"{tag}.lr"] = np.float64(runner.optimizer.param_groups[0]["lr"])
        out[f"{tag}.dis_lr"] = np.float64(runner.dis_optimizer.param_groups[0]["lr"])
        out[f"{tag}.global_step"] = np.int64(loop.global_step)
        out[f"{tag}.finetune"] = np.int64(int(loop.finetune))
        out[f"{tag}.decoder_requires_grad"] = np.int64(int(all(p.requires_grad for p in model.decoder.parameters())))
        out[f"{tag}.disc_requires_grad"] = np.int64(int(any(p.requires_grad for p in disc.parameters())))

    run_epoch, dis_train = loop.run_epoch, loop.Discriminator_train

    def run_epoch_rec():
        run_epoch()
        snap(f"epoch{loop._cur_epoch}")

    def dis_train_rec():
        dis_train()
        snap(f"dis{loop._cur_epoch}")

    loop.run_epoch, loop.Discriminator_train = run_epoch_rec, dis_train_rec
    loop.run()
    assert events == ["dis0", "epoch0", "epoch1", "dis2", "epoch2", "epoch3", "epoch4", "epoch5"], events
    # every batch logs twice: _process_batch's own line (global_step before its increment) and, at log_interval epochs, run_epoch's; keep the former
    per_batch = [float(s.split(":")[-1]) for s in losses]
    per_batch = per_batch[0:2 * nb:2] + per_batch[2 * nb:]             # epoch 0 (0 % log_interval == 0) logs every batch twice
    assert len(per_batch) == 6 * nb
    out["events"] = np.array(events)
    out["loss_strings"] = np.array(losses)
    out["losses"] = np.array(per_batch, np.float64)
    save("g18_train_schedule", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g3b", "g4", "g5", "g6", "g6b", "g7", "g8", "g9", "g9b", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18"]
    for w in which:
        globals()[w]()
