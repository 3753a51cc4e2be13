"""Student + EMA-teacher container -- host-side mirror of models/uscod.py::baseline (:9-22):
``baseline(cfg)(batched_inputs, ema=False)`` -> ``(fg, bg, extra_loss)``, or ``fg`` under no_grad when ``ema``."""
import torch
from torch import nn

from .modules.DBA import RevDecoder
from ..engine.registry import MODULE_REGISTRY


@MODULE_REGISTRY.register()
class baseline(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.decoder = RevDecoder(cfg)
        self.decoder_ema = RevDecoder(cfg, ema=True)

    def forward(self, batched_inputs, ema: bool = False):
        if ema:
            with torch.no_grad():
                return self.decoder_ema(batched_inputs)
        return self.decoder(batched_inputs)
