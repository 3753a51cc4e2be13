"""Deterministic weights for the SparseRefiner fixture (G9), shared by make_golden.py (applied to the REFERENCE module)
and the tests (applied to the build's mirror).  Seeded default init (both modules build identical nn sub-modules in the
same order, so they consume the RNG identically) plus a closed-form perturbation so that the zero-initialised biases of
nn.MultiheadAttention and the LayerNorm affine terms take part in the check."""
import torch

SEED = 9


def perturb_(module):
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            n = p.numel()
            t = torch.arange(n, dtype=torch.float32)
            p.add_((0.02 * torch.sin(0.37 * t + i)).reshape(p.shape))
    return module


def checksums(module):
    return {name: torch.stack([p.detach().double().sum(), p.detach().double().abs().sum()]) for name, p in module.named_parameters()}


def make_inputs(partial):
    g = torch.Generator().manual_seed(90 + int(partial))
    B, H = 2, 6
    l = torch.randn(B, 768, H, H, generator=g)
    h = torch.randn(B, 9, 768, H, H, generator=g)
    preds = torch.randn(B, 1, H, H, generator=g) * 2
    if partial:                                    # confident (low-entropy) regions: only some windows pass the threshold
        preds[0, :, :4, :] = 14.0
        preds[1, :, :, 2:] = -14.0
    return l, h, preds


def make_h_targets(kind, B=2, ws=3, H=6):
    """Per-window high-resolution targets [B*ws*ws, 1, H, H] for the training-mode fixture (G9b): soft masks in [0, 1] ("prob") or logits
    ("logit": values beyond 1, so that binary_iou applies its sigmoid)."""
    g = torch.Generator().manual_seed(95 + (kind == "logit"))
    t = torch.rand(B * ws * ws, 1, H, H, generator=g)
    t[::2, :, :3] = (t[::2, :, :3] > 0.4).float()            # a mix of hard and soft pixels
    return t if kind == "prob" else (t - 0.5) * 9.0


def coral_inputs():
    """Seeded inputs of the CORAL validation-loop vectors (G15): l [1,768,5,5], m [1,4,768,36,36] (the 2x2 overlapping crops of a
    54x54 map, lr_dataset.py:155-166), h [1,9,768,5,5]."""
    g = torch.Generator().manual_seed(150)
    l = torch.randn(1, 768, 5, 5, generator=g)
    full = torch.randn(1, 768, 54, 54, generator=g) * 0.7
    m = torch.stack([full[:, :, i * 18:i * 18 + 36, j * 18:j * 18 + 36] for i in range(2) for j in range(2)], dim=1)
    h = torch.randn(1, 9, 768, 5, 5, generator=g)
    return l, m, h
