"""Pseudo-label generation -- the package's counterpart of the reference's generate_pseudo_label.py script.

``refine_post_process`` keeps the reference's name, signature and quirks (:30-68); ``PseudoLabelGenerator`` replaces the
per-image ``generate_mask`` (:71-94) by a batched pass: HIP backbone -> key map + CLS attention row of the last layer ->
``ucod_bkg_seg`` -> ``1 - bkg_mask`` -> post-process, and ``build_cache`` writes the ``pseudo_label_cache/<DATASET>`` directory
the training set reads (cache_manager.py:61-66; items are ``[1, h, w]`` float tensors like the reference's)."""
import numpy as np
import torch

from .data.utils.found_bkg_mask import bkg_seg_from_key_map
from .engine.runner.loop_look_twice import connected_components


def refine_post_process(mask, area_threshold=4):
    """mask [1,H,W] float {0,1} (CPU) -> [1,H,W] float: 8-connected components smaller than ``area_threshold`` whose one-pixel ring
    is uniformly the opposite of the label sampled at the component's bounding-box centre are flipped (:30-68)."""
    m = mask.detach().cpu().numpy().astype(np.uint8).squeeze()
    n, labels = connected_components(m)
    out = m.copy()
    for lab in range(1, n):
        ys, xs = np.nonzero(labels == lab)
        if len(ys) >= area_threshold:
            continue
        x, y = int(xs.min()), int(ys.min())
        w, h = int(xs.max()) - x + 1, int(ys.max()) - y + 1
        x0, y0 = max(x - 1, 0), max(y - 1, 0)
        x1, y1 = min(x + w + 1, m.shape[1]), min(y + h + 1, m.shape[0])
        ring = np.ones((y1 - y0, x1 - x0), bool)
        ring[ys - y0, xs - x0] = False
        sampled = out[y + h // 2, x + w // 2]
        if np.all(out[y0:y1, x0:x1][ring] == 1 - sampled):
            out[ys, xs] = 1 - sampled
    return torch.tensor(out).unsqueeze(0).float()


class PseudoLabelGenerator:
    def __init__(self, feature_extractor, th_bkg=0.6, area_threshold=4, precision="f32eq"):
        """``precision``: the reference's script runs the backbone in plain fp32 under ``torch.no_grad()`` (generate_pseudo_label.py:71-89: no autocast), and its
        output is a THRESHOLDED map -- so a ``backbone`` wrapper is asked for its f32-equivalent sibling by default (``with_precision("f32eq")``: the
        split-operand engine); ``precision=None`` keeps the extractor as given (a bare engine is always used as it is)."""
        if precision is not None and hasattr(feature_extractor, "with_precision"):
            feature_extractor = feature_extractor.with_precision(precision)
        self.engine = getattr(feature_extractor, "engine", feature_extractor)
        self.th_bkg, self.area_threshold = th_bkg, area_threshold

    @torch.no_grad()
    def raw_masks(self, images):
        """images [B,3,H,W] (already transformed, :116-120) -> 1 - bkg_mask, float [B,h,w] on the device."""
        key, att = self.engine.forward_with_cls_attention(images.to(self.engine.device))
        return 1.0 - bkg_seg_from_key_map(att, key, self.th_bkg)["bkg_mask"]

    def generate_masks(self, images):
        """-> list of [1,h,w] float CPU tensors, one per image (what generate_mask returns, :94)."""
        raw = self.raw_masks(images).cpu()
        return [refine_post_process(m.unsqueeze(0), self.area_threshold) for m in raw]

    def build_cache(self, images, pseudo_label_cache, batch_size=32):
        """Stream an iterable of [3,H,W] tensors into ``pseudo_label_cache`` (a CacheManager in write mode)."""
        store = pseudo_label_cache.io._store
        if store.readable:
            raise RuntimeError(f"cache at {store.dir} already exists and is valid; remove it to rebuild")
        n, pending = 0, []

        def flush():
            nonlocal n
            if pending:
                for m in self.generate_masks(torch.stack(pending)):
                    store.store(n, m)
                    n += 1
                pending.clear()

        for img in images:
            if pending and tuple(img.shape) != tuple(pending[0].shape):
                flush()
            pending.append(img)
            if len(pending) == batch_size:
                flush()
        flush()
        store.commit()
        pseudo_label_cache.io.reload_path()
        return n
