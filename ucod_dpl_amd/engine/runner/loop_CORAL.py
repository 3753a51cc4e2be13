"""CORAL second-stage validation -- host-side mirror of engine/runner/loop_CORAL.py::LocalRefineValidationLoop (:41-341) around the
HIP ``SparseRefiner`` (rows R1-R4) and the HIP decoder.

Same method names and semantics: ``concate_preds`` (:61-95), ``_prepare_validation_features`` (:205-246), ``_should_crop_center``
(:248-259), ``_center_pad`` (:168-203), ``crop_center`` (:276-311), ``process_preds`` (:313-341), ``_process_validation_batch``
(:130-166), ``run`` (:97-128).  Every resize is the HIP bilinear kernel (ATen semantics), the first-stage logits come from the HIP
decoder, the refinement from the HIP refiner; predictions are not written to disk (``_save_prediction_image`` is I/O, out of
scope); the statistics are the nine COD measures of engine/utils/metrics/metric.py::statistics, computed on the device
(``engine/utils/metrics``, csrc/cod_metrics.hip).  ``WindowFeatures`` is the data side (data/datasets/lr_dataset.py:82-166): the 3x3 window key
features and the 2x2 overlapping crops of the 54x54 key map, with every backbone call of an image batched into one pass.
"""
import numpy as np
import torch

from ... import ops, parallel
from .loop_UCOD_DPL import BaseLoop
from .loop_look_twice import MAEStatistics  # noqa: F401  (re-exported)
from ..utils.metrics import statistics


class WindowFeatures:
    """lr_dataset.py::get_features (:82-166) for an in-memory uint8 RGB image."""

    def __init__(self, feature_extractor, look_twice_loop, window_size=3, grid=(518, 518), extractor_size=(756, 756), image_size=(518, 518), precision="f32eq"):
        # the reference computes these features inside its dataset (lr_dataset.py:97-157: a backbone of its own, outside accelerate's autocast) -- plain fp32: a
        # `backbone` wrapper is asked for its f32-equivalent sibling by default (precision=None keeps the extractor as given)
        if precision is not None and hasattr(feature_extractor, "with_precision"):
            feature_extractor = feature_extractor.with_precision(precision)
        self.fe = feature_extractor
        self.lt = look_twice_loop                      # owns the Pillow-exact crop / resize / normalise kernel
        self.window_size, self.grid = window_size, grid
        self.extractor_size, self.image_size = extractor_size, image_size

    def _resized(self, img_u8, out_hw):
        H, W = img_u8.shape[:2]
        saved = self.lt.img_size
        self.lt.img_size = out_hw
        try:
            return self.lt.crop_batch(img_u8, [[0, 0, W, H]])                   # Pillow-BILINEAR resize + ToTensor + Normalize
        finally:
            self.lt.img_size = saved

    def crop_center(self, img_u8):
        """:121-131 (the centred half-size crop)."""
        H, W = img_u8.shape[:2]
        nw, nh = W // 2, H // 2
        left, top = (W - nw) // 2, (H - nh) // 2
        return np.ascontiguousarray(img_u8[top:top + nh, left:left + nw])

    @torch.no_grad()
    def get_features(self, img_u8, require_m_patches=False, crop_center=False):
        """-> (l_features [1,C,h,w] | None, h_inputs [1,ws*ws,C,gh,gw], m_inputs [1,4,C,36,36] | None)."""
        if crop_center:
            img_u8 = self.crop_center(img_u8)
        ws, (gh, gw) = self.window_size, self.grid
        big = self._resized(img_u8, (ws * gh, ws * gw))                         # resize once (:103), crop windows (:137-147): pure slicing
        wins = big[0].unfold(1, gh, gh).unfold(2, gw, gw)                       # [3, ws, ws, gh, gw]
        wins = wins.permute(1, 2, 0, 3, 4).reshape(ws * ws, 3, gh, gw).contiguous()
        _, hk = self.fe(wins)
        h_inputs = hk.unsqueeze(0)
        m_inputs = None
        if require_m_patches:
            _, key = self.fe(self._resized(img_u8, self.extractor_size))        # :151-153, 756 -> 54x54 key map
            m_inputs = torch.stack([key[:, :, i * 18:i * 18 + 36, j * 18:j * 18 + 36] for i in range(2) for j in range(2)], dim=1)
        l_features = None
        if crop_center:
            _, l_features = self.fe(self._resized(img_u8, self.image_size))     # :113-116
        return l_features, h_inputs, m_inputs


class LocalRefineValidationLoop(BaseLoop):
    def __init__(self, config, runner, window_features=None):
        super().__init__(config, runner)
        self._mode = "val"
        self.window_length = self.cfg.model_cfg.window_length
        self.window_features = window_features
        self.device = runner.device

    # ---- tensor helpers (same arithmetic as the reference)
    def concate_preds(self, preds):
        b, n, c, h, w = preds.shape
        full = torch.zeros(b, c, 102, 102, device=preds.device)
        cnt = torch.zeros(b, c, 102, 102, device=preds.device)
        for i in range(2):
            for j in range(2):
                full[:, :, i * 34:i * 34 + 68, j * 34:j * 34 + 68] += preds[:, i * 2 + j]
                cnt[:, :, i * 34:i * 34 + 68, j * 34:j * 34 + 68] += 1.0
        return full / (cnt + 1e-6)

    def _center_pad(self, x, fill_value=-10.0):
        if x.dim() not in (3, 4):
            raise ValueError("shape error:{}".format(x.shape))
        *lead, h, w = x.shape
        out = torch.full((*lead, 2 * h, 2 * w), fill_value, device=x.device, dtype=x.dtype)
        out[..., h // 2:h // 2 + h, w // 2:w // 2 + w] = x
        return out

    def _should_crop_center(self, preds):
        return bool((preds > 0).sum() / (preds.shape[2] * preds.shape[3]) < 0.001)

    def _features_and_preds(self, l_in, m_in, h_in):
        wl = self.window_length
        b, c = l_in.shape[:2]
        l = ops.bilinear_resize(l_in.to(self.device, torch.float32), wl, wl)
        h = ops.bilinear_resize(h_in.to(self.device, torch.float32).flatten(0, 1), wl, wl).reshape(b, -1, c, wl, wl)
        if self.cfg.dataset_cfg.valset_cfg.require_m_patches:
            m = ops.bilinear_resize(m_in.to(self.device, torch.float32).flatten(0, 1), 68, 68)
            preds = self.concate_preds(self.runner.model(m)[0].reshape(b, -1, 1, 68, 68))
        else:
            preds = self.runner.model(l)[0]
        return dict(l_features=l, h_features=h, preds=preds)

    def _prepare_validation_features(self, l_input_features, m_input_features, h_input_features):
        with torch.no_grad():
            return self._features_and_preds(l_input_features, m_input_features, h_input_features)

    def crop_center(self, image):
        """:276-311; ``image`` is a path or a uint8 [H,W,3] array."""
        if isinstance(image, str):
            from PIL import Image
            image = np.asarray(Image.open(image).convert("RGB"))
        l, h, m = self.window_features.get_features(image, self.cfg.dataset_cfg.valset_cfg.require_m_patches, crop_center=True)
        return self._prepare_validation_features(l, m, h)

    def process_preds(self, preds, size):
        h, w = size
        probs = preds if bool(torch.all((preds >= 0) & (preds <= 1))) else torch.sigmoid(preds)
        up = ops.bilinear_resize(probs.to(self.device, torch.float32).contiguous(), h, w)[..., :h, :w]
        return (up > 0.5).squeeze(0).float()

    # ---- the loop
    def _process_validation_batch(self, batch, statistics_val):
        _, label_tensor, l_in, img_path, m_in, h_in, _ = batch.values()
        with torch.no_grad():
            fd = self._prepare_validation_features(l_in, m_in, h_in)
            crop = self._should_crop_center(fd["preds"])
            if crop:
                fd = self.crop_center(img_path[0])
            outputs, _, opt = self.runner.refiner(fd["l_features"], fd["h_features"], fd["preds"])
            if crop:
                outputs = self._center_pad(outputs)
            preds_up = self.process_preds(outputs, tuple(label_tensor.shape[2:]))
            statistics_val.step(label_tensor.to(preds_up.device) if isinstance(statistics_val, statistics) else label_tensor, preds_up)
        return preds_up

    def run(self):
        stats = statistics()                                  # the nine COD measures on the device (engine/utils/metrics/metric.py:19-74)
        self.runner.refiner.eval()
        for batch in parallel.shard(self.runner.val_dataloader):
            self._process_validation_batch(batch, stats)
        stats.gather_records(device=self.device, dataset_len=parallel.padded_sampler_len(self.runner.val_dataloader))
        result = stats.get_result()
        self.runner.logger.log_table({k: [round(v, 4)] for k, v in result.items()})
        return result
