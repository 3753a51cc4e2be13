"""Oracle: CORAL SparseRefiner forward (second stage, inference).  TEST INFRASTRUCTURE ONLY.

Restates models/UDLR.py:77-86 (SparseRefiner.forward, eval mode: cal_ex_loss returns 0), models/modules/ASR.py:41-51
(EntropySelector), models/modules/CSF.py:38-43 + models/modules/mlp.py:134-148 (CrossAttentionBlock around
nn.MultiheadAttention: 8 heads, head_dim 96, LayerNorm eps 1e-5), models/modules/HRE.py:18-39 (window scatter-average)
and models/modules/GE_pix_level.py:16-25 (gated ensembling; note ``en_local.max()`` is over the WHOLE batch).
State-dict keys are the reference's (``HRE.CSF.attn.*``, ``HRE.CSF.depthwise_conv.*``, ``HRE.CSF.mask_dec.*``, ``GE.*``).
Pinned by tests/golden/g9_refiner.npz (the reference module run in eval mode on seeded inputs and seeded default init).
"""
import torch
import torch.nn.functional as F

from .vit import layer_norm, gelu_erf
from .resize import torch_bilinear

HEADS = 8
LN_EPS = 1e-5


def entropy_select(preds, window_size, threshold):
    """ASR.py:41-51.  preds [B,1,H,W] -> (entropy [B,1,H,W], mask bool [B,1,ws,ws], coords [Nw,2] (y,x), counts per image)."""
    probs = preds if bool(torch.all((preds >= 0) & (preds <= 1))) else torch.sigmoid(preds)
    entropy = -probs * torch.log(probs.clamp(1e-5))
    scores = F.adaptive_avg_pool2d(entropy.float(), (window_size, window_size))
    mask = scores > threshold
    coords = []
    for b in range(mask.shape[0]):
        for idx in torch.nonzero(mask[b].flatten()).flatten().tolist():
            coords.append([idx // window_size, idx % window_size])
    return entropy, mask, torch.tensor(coords, dtype=torch.long).reshape(-1, 2), [int(m.sum()) for m in mask]


def cross_attention_block(query, context, sd, p="HRE.CSF.attn."):
    """mlp.py:134-148.  query [N,Q,C], context [N,K,C]."""
    C = query.shape[-1]
    hd = C // HEADS
    q = layer_norm(query, sd[p + "norm_q.weight"], sd[p + "norm_q.bias"], LN_EPS)
    kv = layer_norm(context, sd[p + "norm_kv.weight"], sd[p + "norm_kv.bias"], LN_EPS)
    W, b = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
    Q = q @ W[:C].t() + b[:C]
    K = kv @ W[C:2 * C].t() + b[C:2 * C]
    V = kv @ W[2 * C:].t() + b[2 * C:]
    N, Lq, _ = Q.shape
    Lk = K.shape[1]
    Qh = Q.view(N, Lq, HEADS, hd).transpose(1, 2)
    Kh = K.view(N, Lk, HEADS, hd).transpose(1, 2)
    Vh = V.view(N, Lk, HEADS, hd).transpose(1, 2)
    P = torch.softmax(torch.matmul(Qh, Kh.transpose(2, 3)) * (hd ** -0.5), dim=-1)
    A = torch.matmul(P, Vh).transpose(1, 2).reshape(N, Lq, C)
    A = A @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
    x = query + A
    h = layer_norm(x, sd[p + "norm_mlp.weight"], sd[p + "norm_mlp.bias"], LN_EPS)
    h = gelu_erf(h @ sd[p + "mlp.0.weight"].t() + sd[p + "mlp.0.bias"])
    return x + (h @ sd[p + "mlp.2.weight"].t() + sd[p + "mlp.2.bias"])


def csf(l_windows, h_windows, sd):
    """CSF.py:38-43.  [Nw,C,H,W] x2 -> window logits [Nw,1,H,W]."""
    Nw, C, H, W = h_windows.shape
    q = h_windows.flatten(2, 3).permute(0, 2, 1)
    ctx = l_windows.flatten(2, 3).permute(0, 2, 1)
    x = cross_attention_block(q, ctx, sd)
    x = x.reshape(Nw, H, W, C).permute(0, 3, 1, 2)
    x = F.conv2d(x, sd["HRE.CSF.depthwise_conv.weight"], sd["HRE.CSF.depthwise_conv.bias"], padding=3, groups=C)
    return F.conv2d(x, sd["HRE.CSF.mask_dec.weight"], sd["HRE.CSF.mask_dec.bias"])


def concat_windows(windows, coords, counts, window_size):
    """HRE.py:18-39: place each window at (y*H, x*W); divide by (count + 1e-6); untouched cells stay 0."""
    Nw, C, H, W = windows.shape
    B = len(counts)
    full = torch.zeros(B, C, H * window_size, W * window_size)
    cnt = torch.zeros(B, 1, H * window_size, W * window_size)
    i = 0
    for b in range(B):
        for _ in range(counts[b]):
            y, x = int(coords[i, 0]) * H, int(coords[i, 1]) * W
            full[b, :, y:y + H, x:x + W] += windows[i]
            cnt[b, :, y:y + H, x:x + W] += 1.0
            i += 1
    return full / (cnt + 1e-6)


def gated_ensembler(l1, l2, sd):
    """GE_pix_level.py:16-25."""
    h, w = l2.shape[-2:]
    l1 = torch_bilinear(l1, h, w)
    p = torch.sigmoid(l1)
    g = p.mean(dim=(1, 2, 3), keepdim=True)
    loc = F.avg_pool2d(p.float(), 19, padding=9, stride=1)
    en = -loc * torch.log(loc.clamp(1e-5))
    en = 1 - en / en.max()
    wgt = (en + g) / 2
    y = l1 * wgt + l2 * (1 - wgt)
    hcat = torch.relu(F.conv2d(y, sd["GE.fuser.0.weight"], sd["GE.fuser.0.bias"]))
    return F.conv2d(hcat, sd["GE.fuser.2.weight"], sd["GE.fuser.2.bias"]), wgt


def sparse_refiner_forward(input_features, h_inputs, preds, sd, window_size=3, threshold=0.0015):
    """UDLR.py:77-86 in eval mode -> (outputs [B,1,ws*H,ws*W], opt dict)."""
    entropy, mask, coords, counts = entropy_select(preds, window_size, threshold)
    B = input_features.shape[0]
    sel = mask.flatten()
    h_sel = h_inputs.flatten(0, 1)[sel]                                       # ASR.py:14-20
    l_sel = torch.repeat_interleave(input_features, torch.tensor(counts), dim=0)
    H, W = h_inputs.shape[-2:]
    if h_sel.shape[0] > 0:
        window_preds = csf(l_sel, h_sel, sd)
    else:
        window_preds = torch.zeros(0, 1, H, W)
    h_preds = concat_windows(window_preds, coords, counts, window_size)
    outputs, ge_w = gated_ensembler(preds, h_preds, sd)
    return outputs, dict(mask=mask, entropy=entropy, h_preds=h_preds, window_preds=window_preds, GE_w=ge_w, preds=preds, coords_list=coords)


def binary_iou(preds, targets, threshold=0.5):
    """UDLR.py:26-42."""
    if preds.ndim == 4:
        preds = preds.squeeze(1)
    if targets.ndim == 4:
        targets = targets.squeeze(1)
    if preds.dtype.is_floating_point:
        preds = torch.sigmoid(preds) if preds.max() > 1 else preds
        preds = (preds > threshold).int()
    targets = targets.int()
    inter = (preds & targets).sum(dim=(1, 2)).float()
    union = (preds | targets).sum(dim=(1, 2)).float()
    return inter / (union + 1e-6)


def cal_ex_loss(preds, window_preds, mask, h_targets, window_size):
    """UDLR.py:52-75 in TRAINING mode -> (loss, window_targets, ious): the IoU-weighted window loss.  ``preds`` [B,1,ph,pw] first-stage
    logits, ``window_preds`` [n,1,h,w], ``mask`` bool [B,1,ws,ws], ``h_targets`` [B*ws*ws,1,h,w]."""
    if mask.sum() == 0:
        return torch.zeros(()), None, None
    _, _, h, w = window_preds.shape
    ws = window_size
    l = F.interpolate(preds, size=(h * ws, w * ws), mode="bilinear").sigmoid() > 0.5
    l = F.unfold(l.float(), kernel_size=(h, w), stride=(h, w)).reshape(-1, 1, h, w, ws ** 2).permute(0, 4, 1, 2, 3).flatten(0, 1)
    l = l[mask.flatten()]
    t = h_targets[mask.flatten()]
    ious = (binary_iou(t, l).view(-1, 1, 1, 1) * 1.5).clamp(0, 1)
    bce = torch.nn.BCEWithLogitsLoss(reduction="none")
    loss = (ious * bce(window_preds, t) + (1 - ious) * bce(window_preds, l)).mean() / 2
    return loss, t, ious.view(-1)
