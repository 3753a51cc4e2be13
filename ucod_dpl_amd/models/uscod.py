"""Student + EMA-teacher container: the ``baseline`` of models/uscod.py (reference :9-22) on the HIP decoder.

Call shape kept: ``baseline(cfg)(features, ema=False) -> (fg, bg, extra_loss)`` for the student, and with ``ema=True`` the
teacher's ``fg`` logits, computed without an autograd graph.  Attribute names ``decoder`` / ``decoder_ema`` are what the
shipped ``weights/UCOD_DPL_*.safetensors`` key on."""
import torch
from torch import nn

from .modules.DBA import RevDecoder
from ..engine.registry import MODULE_REGISTRY


@MODULE_REGISTRY.register()
class baseline(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        branches = {"decoder": RevDecoder(cfg), "decoder_ema": RevDecoder(cfg, ema=True)}
        for name, module in branches.items():
            self.add_module(name, module)

    def branch(self, ema: bool):
        return self.decoder_ema if ema else self.decoder

    def forward(self, batched_inputs, ema: bool = False):
        with torch.set_grad_enabled(torch.is_grad_enabled() and not ema):
            return self.branch(ema)(batched_inputs)
