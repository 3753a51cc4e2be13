"""Oracle: image/feature resampling semantics.  TEST INFRASTRUCTURE ONLY.

The reference calls three different resamplers on the hot path; none of them is vendored
in /root/reference, so their published algorithms are restated here and pinned against
the installed libraries (torch 2.10 / Pillow 12.2) in tests/test_oracle_resize.py:

* ``torch_bilinear``  -- F.interpolate(mode='bilinear', align_corners=False), no antialias
  (loop_UCOD_DPL.py:153-154,236,241,305,315,356-358).  ATen upsample_bilinear2d.
* ``torch_bicubic``   -- F.interpolate(mode='bicubic', align_corners=False), A=-0.75
  (HF modeling_dinov2.py:85-92 ``size=``; models/backbones/dino.py:217-221 ``scale_factor=``).
* ``pil_resize_u8``   -- Pillow ImagingResample on 8-bit images (antialiased separable
  filter with 22-bit fixed-point coefficients): torchvision ``Resize`` on a PIL image
  (loop_UCOD_DPL.py:283,342; BILINEAR) and ``Image.resize`` default BICUBIC on the 'L'
  mask (loop_UCOD_DPL.py:350).
"""
import math
import numpy as np
import torch


# --------------------------------------------------------------------------- torch semantics
def _src_index(out_size, in_size, scale, cubic):
    """area_pixel_compute_source_index: ``scale * (dst + 0.5) - 0.5`` in fp32.  Both the ATen CPU
    build (-mfma) and the GPU compilers contract this into ONE fused multiply-add, which matters
    when a pixel lands exactly between two source pixels (lambda = 0.5 ties under a >0.5 threshold,
    loop_UCOD_DPL.py:241,261): emulate the single rounding by evaluating in fp64 (exact here)."""
    dst = torch.arange(out_size, dtype=torch.float64)
    s32 = torch.tensor(scale, dtype=torch.float32).double()
    src = (s32 * (dst + 0.5) - 0.5).float()
    if not cubic:
        src = src.clamp_min(0.0)                       # area_pixel_compute_source_index
    return src


def torch_bilinear(x, oh, ow):
    """x [..., H, W] float -> [..., oh, ow]; ATen upsample_bilinear2d, align_corners=False."""
    H, W = x.shape[-2:]
    sy = _src_index(oh, H, H / oh, False)
    sx = _src_index(ow, W, W / ow, False)
    y0 = sy.floor().long().clamp_max(H - 1)
    x0 = sx.floor().long().clamp_max(W - 1)
    y1 = (y0 + 1).clamp_max(H - 1)
    x1 = (x0 + 1).clamp_max(W - 1)
    ly = (sy - y0.float()).to(x.dtype)
    lx = (sx - x0.float()).to(x.dtype)
    top = x[..., y0, :]
    bot = x[..., y1, :]

    def horiz(r):
        return r[..., x0] * (1 - lx) + r[..., x1] * lx

    return horiz(top) * (1 - ly).unsqueeze(-1) + horiz(bot) * ly.unsqueeze(-1)


def _cubic_coeffs(t, A=-0.75):
    def c1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return torch.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], -1)


def torch_bicubic(x, oh, ow, scale_h=None, scale_w=None):
    """x [..., H, W]; ``scale_*`` = the scale_factor the caller passed (then ATen uses
    1/scale_factor as the coordinate scale), else in/out."""
    H, W = x.shape[-2:]
    sh = (1.0 / scale_h) if scale_h else H / oh
    sw = (1.0 / scale_w) if scale_w else W / ow
    sy = _src_index(oh, H, sh, True)
    sx = _src_index(ow, W, sw, True)
    iy, ix = sy.floor(), sx.floor()
    cy = _cubic_coeffs(sy - iy).to(x.dtype)            # [oh,4]
    cx = _cubic_coeffs(sx - ix).to(x.dtype)
    out = torch.zeros(*x.shape[:-2], oh, ow, dtype=x.dtype)
    for a in range(4):
        yy = (iy.long() - 1 + a).clamp(0, H - 1)
        row = x[..., yy, :]
        acc = torch.zeros(*x.shape[:-2], oh, ow, dtype=x.dtype)
        for b in range(4):
            xx = (ix.long() - 1 + b).clamp(0, W - 1)
            acc = acc + row[..., xx] * cx[:, b]
        out = out + acc * cy[:, a].unsqueeze(-1)
    return out


# --------------------------------------------------------------------------- Pillow semantics
PRECISION_BITS = 32 - 8 - 2


def _bilinear_filter(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic_filter(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


PIL_FILTERS = {"bilinear": (_bilinear_filter, 1.0), "bicubic": (_bicubic_filter, 2.0)}


def pil_coeffs(in_size, out_size, filt, in0=0.0, in1=None):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out,ksize])."""
    f, support0 = PIL_FILTERS[filt]
    in1 = float(in_size) if in1 is None else in1
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis_u8(img, out_size, filt, axis, in0=0.0, in1=None):
    """One Pillow 8bpc pass along ``axis`` of an [H,W,C] uint8 array."""
    img = np.moveaxis(img, axis, 0)
    bounds, kk = pil_coeffs(img.shape[0], out_size, filt, in0, in1)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_u8(img, out_w, out_h, filt, box=None):
    """Pillow Image.resize((out_w,out_h), filt, box) on uint8 [H,W] or [H,W,C].
    ``box`` = (left, top, right, bottom) in source pixels (floats allowed).  Horizontal pass
    first, then vertical, intermediate rounded to uint8 (Resample.c ImagingResampleInner)."""
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W = img.shape[:2]
    l, t, r, b = box if box is not None else (0.0, 0.0, float(W), float(H))
    need_h = out_w != W or l != 0 or r != W
    need_v = out_h != H or t != 0 or b != H
    cur = img
    if need_h:
        if need_v:
            # Pillow only resamples the rows the vertical pass will touch; rows outside are never read,
            # so resampling all rows and letting the vertical pass index them is equivalent.
            pass
        cur = _resample_axis_u8(cur, out_w, filt, 1, l, r)
    if need_v:
        cur = _resample_axis_u8(cur, out_h, filt, 0, t, b)
    if not need_h and not need_v:
        cur = img.copy()
    return cur[:, :, 0] if squeeze else cur
