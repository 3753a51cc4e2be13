"""LoRA-backbone end-to-end model -- host-side mirror of models/modules/full_model.py (the reference file itself cannot be
imported: it needs ``peft`` and a module, ``models/modules/ocm.py``, that is not in the repository).

Same names and call shapes: ``load_lora(cfg, model)``, ``get_full_model(cfg, checkpoint_path)``,
``full_model(config, backbone, decoder)(inputs, ema=False, get_hidden_feature=False)``, ``freeze_lora`` / ``active_lora``,
``load_state_dict`` (decoder only, :146-147).  What runs underneath is the HIP training engine (``ViTLoRAEngine``):
LoRA r / lora_alpha / target modules as :47-72 (query, key, value; bias 'none'), the key hook of the last layer -> CLS
dropped -> NCHW -> bilinear 68x68 (:95-106), student backbone differentiable w.r.t. its LoRA matrices, EMA backbone
frozen.  ``enable_ocm`` is rejected: its module does not exist in the reference.  LoRA dropout (``lora_dropout``, 0.05 in the
reference config) is applied in train mode with counter-based masks (it cannot reproduce torch's RNG stream).
"""
import os

import torch
from torch import nn

from ... import ops
from ...vit_engine import ViTLoRAEngine
from ..uscod import baseline


class LoRABackbone(nn.Module):
    """nn.Module face of a ViTLoRAEngine: ONE flat parameter ``lora`` [L, 6*r*D] that aliases the engine's arena."""

    def __init__(self, engine: ViTLoRAEngine):
        super().__init__()
        self.engine = engine
        self.lora = nn.Parameter(engine.lora, requires_grad=True)          # same storage: optimiser steps are seen by the engine

    def forward(self, pixel_values):
        if self.lora.requires_grad and torch.is_grad_enabled():
            return self.engine.apply(pixel_values, self.lora)
        return self.engine.forward_train(pixel_values)

    def train(self, mode=True):
        self.engine.train(mode)                                    # LoRA dropout follows the module's mode, like peft's nn.Dropout
        return super().train(mode)

    def sync(self):
        """Re-derive the LoRA columns of the augmented GEMM weights after ``lora`` changed (optimiser step / EMA / load)."""
        self.engine.repack()


def load_lora(config, state_dict, heads, device="cuda", generator=None):
    """models/modules/full_model.py:47-72.  r == 0 is refused (the reference returns the bare model; use ``backbone`` then)."""
    r = getattr(config, "r", 2)
    if r == 0:
        raise ValueError("r == 0: no LoRA -- use data.utils.feature_extractor.backbone for the frozen path")
    alpha = getattr(config, "lora_alpha", 4)
    targets = list(getattr(config, "target_modules", ["query", "value", "key"]))
    if sorted(targets) != ["key", "query", "value"]:
        raise NotImplementedError(f"target_modules {targets}: only query/key/value (the reference default) is built")
    if getattr(config, "bias", "none") != "none":
        raise NotImplementedError("LoRA bias modes other than 'none' are not built")
    drop = float(getattr(config, "lora_dropout", 0.05))                      # :50
    return LoRABackbone(ViTLoRAEngine(state_dict, heads, r=r, lora_alpha=alpha, device=device, generator=generator, lora_dropout=drop))


class full_model(nn.Module):
    def __init__(self, config, backbone: LoRABackbone, decoder: baseline):
        super().__init__()
        self.config = config
        self.enable_ocm = bool(getattr(config.model_cfg, "enable_ocm", False))
        if self.enable_ocm:
            raise NotImplementedError("enable_ocm: models/modules/ocm.py is not part of the reference repository")
        self.backbone = backbone
        self.backbone_ema = LoRABackbone(backbone.engine.clone_for_ema())       # :84 copy.deepcopy(backbone)
        self.freeze_model(self.backbone_ema)
        self.decoder = decoder
        if getattr(config.model_cfg, "freeze_lora", False):
            self.freeze_lora()
        self.key = None
        self.hook_size = 68                                                      # :103 ih = iw = 68

    def hook_fn_key(self, key_map):
        """:95-106 -- the engine already returns the key projection with CLS dropped as [B,C,h,w]; bilinear to 68x68."""
        if key_map.requires_grad:
            self.key = ops.bilinear_resize_autograd(key_map, self.hook_size, self.hook_size)
        else:
            self.key = ops.bilinear_resize(key_map, self.hook_size, self.hook_size)
        return self.key

    def forward(self, inputs, ema=False, get_hidden_feature=False):
        if ema:
            with torch.no_grad():
                self.hook_fn_key(self.backbone_ema(inputs))
        else:
            self.hook_fn_key(self.backbone(inputs))
        if get_hidden_feature:
            return self.key
        if not ema:
            return self.decoder(self.key, ema=False)                             # (preds, preds_rev, extra_loss)
        return self.decoder(self.key, ema=True)

    def freeze_model(self, model, exclude_keywords=None):
        exclude_keywords = exclude_keywords or []
        for name, param in model.named_parameters():
            param.requires_grad = any(k in name.lower() for k in exclude_keywords)

    def freeze_lora(self):
        for name, param in self.backbone.named_parameters():
            if "lora" in name:
                param.requires_grad = False

    def active_lora(self):
        for name, param in self.backbone.named_parameters():
            if "lora" in name:
                param.requires_grad = True

    def load_state_dict(self, state_dict, strict=True):
        return self.decoder.load_state_dict(state_dict, strict=strict)


def get_full_model(cfg, backbone_state_dict, heads, checkpoint_path=None, device="cuda"):
    """:25-37 with the backbone weights passed in (the reference downloads them through build_feature_extractor)."""
    model = baseline(cfg.model_cfg).to(device)
    if checkpoint_path and os.path.isfile(checkpoint_path):
        from safetensors.torch import load_file
        model.load_state_dict(load_file(checkpoint_path, device=str(device)))
    fe = load_lora(getattr(cfg, "lora_cfg", object()), backbone_state_dict, heads, device=device)
    return full_model(cfg, fe, model)
