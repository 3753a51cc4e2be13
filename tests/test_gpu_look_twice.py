"""GPU: batched crop+resize kernel (bit-identical to Pillow via the oracle) and the whole Look-Twice refinement."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from oracle import look_twice as OLT  # noqa: E402
from ucod_dpl_amd.engine.config import CfgNode  # noqa: E402
from ucod_dpl_amd.engine.runner import loop_look_twice as LT  # noqa: E402
from ucod_dpl_amd.data.utils.feature_extractor import backbone, random_state_dict, ARCHS  # noqa: E402
from ucod_dpl_amd.models.uscod import baseline  # noqa: E402


def synthetic_image(H=427, W=640, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([(xx * 255 / W), (yy * 255 / H), ((xx + yy) % 256)], -1).astype(np.float32)
    img += rng.normal(0, 20, img.shape)
    for _ in range(6):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(10, 60)
        img[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] += rng.integers(-120, 120)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_loop(img_size=(518, 518), th=0.15):
    ARCHS["lt_vit"] = (128, 2, 2, 14, 518, True)
    dev = torch.device("cuda", 0)
    bb = backbone.from_state_dict(random_state_dict("lt_vit", seed=3, image_size=img_size[0]), heads=2, device=dev)
    torch.manual_seed(5)
    model = baseline(CfgNode(dict(dim=128, feature_size=68, ema_weight=0.99, dis_use_features=False))).to(dev)
    runner = types.SimpleNamespace(device=dev, model=model, world_size=1, rank=0, val_dataloader=[], logger=None)
    cfg = CfgNode(dict(train_cfg=dict(dist_train=False), model_cfg=dict(feature_size=68),
                       val_cfg=dict(look_twice=True, look_twice_th=th, expand_type="dynamic"),
                       dataset_cfg=dict(valset_cfg=dict(image_size=img_size))))
    return LT.ValLoop_Look_Twice(cfg, runner, feature_extractor=bb), bb, model


def test_crop_resize_norm_is_bit_identical_to_pillow_path():
    loop, _, _ = make_loop()
    img = synthetic_image()
    boxes = [[100, 50, 300, 200], [0, 0, 640, 427], [600, 400, 100, 80], [-20, -10, 90, 70], [10, 10, 37, 518], [320, 200, 1200, 900]]
    out = loop.crop_batch(img, boxes).cpu()
    for b, o in zip(boxes, out):
        ref = OLT.crop_resize_normalize(img, b, (518, 518))
        assert torch.equal(o, ref), (b, (o - ref).abs().max())


def test_process_preds_boxes_match_oracle():
    loop, _, _ = make_loop()
    g = torch.Generator().manual_seed(11)
    for i in range(6):
        logits = torch.randn(1, 1, 68, 68, generator=g) * 0.5 - 2.0
        for _ in range(i % 4):
            cy, cx, r = (int(v) for v in torch.randint(8, 60, (3,), generator=g))
            r = 2 + r % 7
            logits[..., max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = 4.0
        up, boxes = loop.process_preds(logits.cuda())
        try:
            ref_up, ref_boxes = OLT.process_preds(logits, (518, 518), 0.15, "dynamic")
        except ValueError:
            continue
        assert torch.equal(up.cpu(), ref_up)
        assert boxes == ref_boxes


def test_look_twice_end_to_end_matches_oracle_composition():
    loop, bb, model = make_loop()
    img = synthetic_image(seed=4)
    old = torch.zeros(1, 518, 518)
    old[:, 200:260, 100:180] = 1.0
    bboxes = [[80, 170, 140, 130], [300, 300, 100, 90], list(LT.DEFAULT_BOX)]

    def encode(crop):                                          # same GPU encoder: isolates crop / resize / paste logic
        with torch.no_grad():
            _, key = bb(crop.cuda())
            return model(key)[0].cpu()

    ref = OLT.look_twice(img, bboxes, old.clone(), (518, 518), encode)
    got = loop.look_twice(img, bboxes, old.clone())
    assert torch.equal(got, ref)
    assert (got != old).any()                                   # something was actually refined
