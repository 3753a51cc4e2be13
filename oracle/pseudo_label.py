"""Oracle: pseudo-label generator.  TEST INFRASTRUCTURE ONLY.

``bkg_seg`` restates data/utils/found_bkg_mask.py:4-86 (FOUND-style background discovery: CroW sparsity weights of the last
layer's CLS attention per head :31-36, weighted + L2-normalised key descriptors :39-55, seed = least-attended patch :58-63,
background = patches whose cosine similarity to the seed exceeds ``th_bkg`` :66-75, similarity map :77-82).  Only the seed's row
of the similarity matrix is formed (the reference builds the full HW x HW product and reads one row).
``refine_post_process`` restates generate_pseudo_label.py:30-68 (small 8-connected components whose 1-pixel ring is uniformly the
opposite label are flipped; note the label is sampled at the bounding-box CENTRE, :58, which need not belong to the component).
Pinned by tests/golden/g14_pseudo_label.npz (both reference functions run on a seeded HF Dinov2Model).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .look_twice import connected_components
from .resize import torch_bilinear


def cls_attention(h_last_ln, sd, layer, heads):
    """outputs.attentions[-1][:, :, 0, :] from the last layer's LN1 output [B,N,D]: softmax over ALL keys (CLS included) of the
    CLS query (modeling_dinov2.py:153-179)."""
    a = f"encoder.layer.{layer}.attention.attention."
    B, N, D = h_last_ln.shape
    hd = D // heads
    q = (h_last_ln[:, 0] @ sd[a + "query.weight"].t() + sd[a + "query.bias"]).view(B, heads, 1, hd)
    k = (h_last_ln @ sd[a + "key.weight"].t() + sd[a + "key.bias"]).view(B, N, heads, hd).transpose(1, 2)
    return torch.softmax((q @ k.transpose(2, 3)) * hd ** -0.5, dim=-1)[:, :, 0, :]


def bkg_seg(att_cls, feats, grid, th_bkg, up_size=None, dim=64, epsilon=1e-10, apply_weights=True):
    """att_cls [B,nh,N] = attentions[:, :, 0, :] (column 0 = CLS key), feats [B,N,C] key features (row 0 = CLS).
    -> (bkg_mask [B,g,g] in {0,1}, sim_map * (1 - bkg_mask))."""
    gw, gh = grid
    up = gw if up_size is None else up_size
    B, nh = att_cls.shape[:2]
    att = att_cls[:, :, 1:].reshape(B, nh, gw, gh)
    att = torch_bilinear(att, up, up)
    thr = att.reshape(B, -1).mean(1)
    Q = (att.reshape(B, nh, -1) > thr[:, None, None]).sum(2) / (up * up)
    beta = torch.log((Q + epsilon).sum(1)[:, None] / (Q + epsilon))
    d = feats[:, 1:].reshape(B, -1, nh, dim)
    if apply_weights:
        d = d * beta[:, None, :, None]
    d = d.reshape(B, gw, gh, nh * dim).permute(0, 3, 1, 2)
    d = torch_bilinear(d, up, up).permute(0, 2, 3, 1).reshape(B, up * up, nh * dim)
    d = F.normalize(d, dim=-1, p=2)
    a = att * beta[:, :, None, None] if apply_weights else att
    ref = a.sum(1).reshape(B, -1).argmin(-1)
    row = torch.einsum("bc,bpc->bp", d[torch.arange(B), ref], d).reshape(B, up, up)
    bkg = row > th_bkg
    sim = 1 - row.float()
    sim = sim / (sim.max() + 1e-10)
    return bkg.float(), (sim * (1 - bkg.float())).float(), row


def refine_post_process(mask, area_threshold=4):
    """mask [1,H,W] float {0,1} -> [1,H,W] float."""
    m = mask.numpy().astype(np.uint8).squeeze()
    n, labels = connected_components(m)
    out = m.copy()
    for lab in range(1, n):
        ys, xs = np.nonzero(labels == lab)
        if len(ys) >= area_threshold:
            continue
        x, y, w, h = xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1
        x0, y0, x1, y1 = max(x - 1, 0), max(y - 1, 0), min(x + w + 1, m.shape[1]), min(y + h + 1, m.shape[0])
        ring = np.ones((y1 - y0, x1 - x0), bool)
        ring[ys - y0, xs - x0] = False
        sampled = out[y + h // 2, x + w // 2]                     # bounding-box centre, as the reference samples it
        if np.all(out[y0:y1, x0:x1][ring] == 1 - sampled):
            out[ys, xs] = 1 - sampled
    return torch.tensor(out).unsqueeze(0).float()
