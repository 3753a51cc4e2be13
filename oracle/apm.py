"""Oracle: Adaptive Pseudo-label Module (APM) fusion and the losses.  TEST INFRASTRUCTURE ONLY.

Restates engine/runner/loop_UCOD_DPL.py:257-272 (merge_pseudo_label) and :161-173
(the two BCE-with-logits terms, ``loss -= dis_loss``, ``+ extra_loss``).
"""
import math
import torch

from .discriminator import discriminator_forward


def bce_with_logits_mean(x, t):
    """nn.BCEWithLogitsLoss() (mean): (1-t)*x + log(1+exp(-|x|)) + max(-x,0)."""
    return ((1 - t) * x + torch.clamp(-x, min=0) + torch.log1p(torch.exp(-x.abs()))).mean()


def bce_mean(p, t):
    """nn.BCELoss() (mean) with torch's log clamp at -100."""
    lp = torch.log(p).clamp_min(-100.0)
    l1p = torch.log(1 - p).clamp_min(-100.0)
    return -(t * lp + (1 - t) * l1p).mean()


def apm_weight(p_s, p_p, cur_epoch, max_epoch, start_finetune):
    """loop_UCOD_DPL.py:266-267.  p_s,p_p [B,1] -> w [B,1] in [0,1]."""
    w = 0.5 * (1 + torch.cos(torch.abs(p_s - p_p) * math.pi)) + cur_epoch / (max_epoch + start_finetune)
    return torch.clamp(w, 0, 1)


def merge_pseudo_label(pseudo_labels, p_teachers, p_students, disc_state, cur_epoch, max_epoch, start_finetune):
    """loop_UCOD_DPL.py:257-272.  All maps [B,1,H,W]; ``disc_state`` running stats are mutated
    by both discriminator calls (student mask first, then thresholded pseudo label).
    Returns (merged, dis_loss, w[B,1], p_s, p_p)."""
    pt = (torch.sigmoid(p_teachers) > 0.5).float()
    ps = (torch.sigmoid(p_students) > 0.5).float()
    p_s = discriminator_forward(ps, disc_state)
    p_p = discriminator_forward((pseudo_labels > 0.5).float(), disc_state)
    w = apm_weight(p_s, p_p, cur_epoch, max_epoch, start_finetune)
    w4 = w.unsqueeze(-1).unsqueeze(-1)
    dis_loss = bce_mean(p_s, torch.zeros_like(p_s))
    return pseudo_labels * (1 - w4) + pt * w4, dis_loss, w, p_s, p_p
