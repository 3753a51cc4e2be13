"""Oracle: CORAL second-stage validation loop pieces.  TEST INFRASTRUCTURE ONLY.

Restates engine/runner/loop_CORAL.py::LocalRefineValidationLoop: ``concate_preds`` :61-95 (2x2 overlapping 68x68 patches at stride
34 averaged into 102x102), ``_prepare_validation_features`` :205-246, ``_should_crop_center`` :248-259, ``_center_pad`` :168-203,
``process_preds`` :313-341.  Pinned by tests/golden/g15_coral_loop.npz (the real class on the real baseline / SparseRefiner).
"""
import torch

from .resize import torch_bilinear
from . import decoder as OD


def concate_preds(preds):
    b, n, c, h, w = preds.shape
    full = torch.zeros(b, c, 102, 102)
    cnt = torch.zeros(b, c, 102, 102)
    for i in range(2):
        for j in range(2):
            full[:, :, i * 34:i * 34 + 68, j * 34:j * 34 + 68] += preds[:, i * 2 + j]
            cnt[:, :, i * 34:i * 34 + 68, j * 34:j * 34 + 68] += 1.0
    return full / (cnt + 1e-6)


def prepare_validation_features(l_in, m_in, h_in, dec_params, window_length, require_m_patches):
    b, c = l_in.shape[:2]
    wl = window_length
    l = torch_bilinear(l_in, wl, wl)
    h = torch_bilinear(h_in.flatten(0, 1), wl, wl).reshape(b, -1, c, wl, wl)
    if require_m_patches:
        m = torch_bilinear(m_in.flatten(0, 1), 68, 68)
        p, _, _ = OD.rev_decoder_forward(m, dec_params, orth="gram")
        preds = concate_preds(p.reshape(b, -1, 1, 68, 68))
    else:
        preds, _, _ = OD.rev_decoder_forward(l, dec_params, orth="gram")
    return dict(l_features=l, h_features=h, preds=preds)


def should_crop_center(preds):
    return bool((preds > 0).sum() / (preds.shape[2] * preds.shape[3]) < 0.001)


def center_pad(x, fill_value=-10.0):
    *lead, h, w = x.shape
    out = torch.full((*lead, 2 * h, 2 * w), fill_value, dtype=x.dtype)
    out[..., h // 2:h // 2 + h, w // 2:w // 2 + w] = x
    return out


def process_preds(preds, size):
    h, w = size
    probs = preds if bool(torch.all((preds >= 0) & (preds <= 1))) else torch.sigmoid(preds)
    up = torch_bilinear(probs, h, w)[..., :h, :w]
    return (up > 0.5).squeeze(0).float()
