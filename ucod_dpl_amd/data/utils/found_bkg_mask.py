"""Background discovery for pseudo labels -- host-side mirror of data/utils/found_bkg_mask.py::compute_img_bkg_seg on the HIP
kernels of csrc/pseudo_label.hip.

Same signature and returns.  ``attentions`` may be the full ``[B, nh, N, N]`` tensor the HF model returns or just its CLS row
(``[B, nh, N]`` / ``[B, nh, 1, N]``): only ``[:, :, 0, 1:]`` is read (:23).  ``bkg_seg_from_key_map`` is the zero-copy entry for
the engine's own outputs (NCHW key map + CLS attention row)."""
import torch

from ... import native as N


def bkg_seg_from_key_map(att, key_map, th_bkg, epsilon=1e-10, apply_weights=True):
    """att f32 [B,nh,hw] (patch columns of the CLS attention row), key_map f32 [B,C,h,w] -> dict(bkg_mask, sim_map [B,h,w], cos_row,
    seed, beta)."""
    if not (att.is_cuda and key_map.is_cuda):
        raise RuntimeError("compute_img_bkg_seg runs on the HIP path only")
    B, C, h, w = key_map.shape
    nh, hw = att.shape[1], h * w
    if C != nh * 64:
        raise ValueError(f"head_dim must be 64 (C={C}, heads={nh})")
    att = att.to(torch.float32).contiguous()
    key_map = key_map.to(torch.float32).contiguous()
    dev = key_map.device
    mask = torch.empty(B, hw, device=dev)
    sim = torch.empty(B, hw, device=dev)
    cos = torch.empty(B, hw, device=dev)
    seed = torch.empty(B, dtype=torch.int32, device=dev)
    beta = torch.empty(B, nh, device=dev)
    scratch = torch.empty(1, dtype=torch.int32, device=dev)
    N.check(N.load().ucod_bkg_seg(N.ptr(att), N.ptr(key_map), float(th_bkg), float(epsilon), int(bool(apply_weights)), N.ptr(mask), N.ptr(sim),
                                  N.ptr(cos), N.ptr(seed), N.ptr(beta), N.ptr(scratch), B, nh, hw, N.stream()), "ucod_bkg_seg")
    return dict(bkg_mask=mask.view(B, h, w), sim_map=sim.view(B, h, w), cos_row=cos.view(B, h, w), seed=seed, beta=beta)


def compute_img_bkg_seg(attentions, feats, featmap_dims, th_bkg, up_size=None, dim=64, epsilon=1e-10, apply_weights=True):
    w_f, h_f = featmap_dims
    if up_size is not None and up_size != w_f:
        raise NotImplementedError("up_size != grid: the generator never uses it (generate_pseudo_label.py:83-89)")
    if dim != 64:
        raise NotImplementedError("head_dim 64 only")
    if attentions.dim() == 4:
        attentions = attentions[:, :, 0, :]
    att = attentions[:, :, 1:]
    B, Ntok, C = feats.shape
    key_map = feats[:, 1:, :].transpose(1, 2).reshape(B, C, w_f, h_f)
    r = bkg_seg_from_key_map(att, key_map, th_bkg, epsilon, apply_weights)
    return r["bkg_mask"], r["sim_map"]
