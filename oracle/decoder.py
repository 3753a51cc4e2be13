"""Oracle: Dual-Branch Adversarial (DBA) decoder.  TEST INFRASTRUCTURE ONLY.

Restates models/modules/DBA.py:31-59 (RevDecoder.forward) and :25-29
(calc_orthogonal_loss) as explicit tensor arithmetic, plus the exact Gram-form
rewrite of the orthogonality loss and the closed-form backward the HIP kernels
implement (checked against autograd of the naive form in tests).

Parameter dict keys follow the reference state_dict:
  decoupling.weight [128,C,1,1]  decoupling.bias [128]  learnable_embedding [2,64]
  conv_out_fg.weight [1,64,1,1]  conv_out_fg.bias [1]   conv_out_bg.* likewise
"""
import torch

EMB = 64
NORM_EPS = 1e-12  # F.normalize default eps, DBA.py:40-41


def init_params(dim, generator=None, dtype=torch.float32):
    """Same distributions as nn.Conv2d / nn.Parameter(torch.randn) in DBA.py:13-18."""
    g = generator

    def conv_init(out_c, in_c):
        bound = 1.0 / (in_c ** 0.5)  # kaiming_uniform(a=sqrt(5)) on 1x1 == U(-1/sqrt(fan_in), ..)
        w = (torch.rand(out_c, in_c, 1, 1, generator=g, dtype=dtype) * 2 - 1) * bound
        b = (torch.rand(out_c, generator=g, dtype=dtype) * 2 - 1) * bound
        return w, b

    p = {}
    p["decoupling.weight"], p["decoupling.bias"] = conv_init(2 * EMB, dim)
    p["learnable_embedding"] = torch.randn(2, EMB, generator=g, dtype=dtype)
    p["conv_out_fg.weight"], p["conv_out_fg.bias"] = conv_init(1, EMB)
    p["conv_out_bg.weight"], p["conv_out_bg.bias"] = conv_init(1, EMB)
    return p


def orth_loss_naive(f1, f2):
    """DBA.py:25-29.  f1,f2: [B,HW,64].  Materialises [B,HW,HW]."""
    dot = torch.bmm(f1, f2.transpose(1, 2))
    eye = torch.eye(f1.size(1), dtype=f1.dtype)
    return ((dot * (1 - eye)).pow(2)).mean()


def orth_loss_gram(f1, f2):
    """Exact O(HW*64^2) rewrite:
    sum_{i!=j} (f1_i . f2_j)^2 = tr(G1 G2) - sum_i (f1_i . f2_i)^2,  G_k = F_k^T F_k."""
    B, HW, _ = f1.shape
    g1 = torch.bmm(f1.transpose(1, 2), f1)
    g2 = torch.bmm(f2.transpose(1, 2), f2)
    s = (f1 * f2).sum(-1)
    tot = (g1 * g2).sum() - (s * s).sum()
    return tot / (B * HW * HW)


def _branches(x, p):
    """DBA.py:35-41: 1x1 decoupling conv, chunk, per-channel scale, L2-normalise over the HW axis."""
    B, C, H, W = x.shape
    w = p["decoupling.weight"].reshape(2 * EMB, C)
    d = torch.einsum("nc,bcp->bnp", w, x.reshape(B, C, H * W)) + p["decoupling.bias"].view(1, -1, 1)
    d1, d2 = d[:, :EMB], d[:, EMB:]                       # [B,64,HW]
    e = p["learnable_embedding"]
    u1 = d1 * e[0].view(1, -1, 1)
    u2 = d2 * e[1].view(1, -1, 1)
    n1 = u1.norm(p=2, dim=2, keepdim=True).clamp_min(NORM_EPS)   # norm over pixels
    n2 = u2.norm(p=2, dim=2, keepdim=True).clamp_min(NORM_EPS)
    return d1, d2, u1 / n1, u2 / n2, n1, n2


def rev_decoder_forward(x, p, ema=False, orth="naive"):
    """Returns (fg[B,1,H,W], bg[B,1,H,W], extra_loss or None)."""
    if isinstance(x, list):
        x = x[-1]                                         # DBA.py:32
    B, C, H, W = x.shape
    d1, d2, f1, f2, _, _ = _branches(x, p)
    extra = None
    if not ema:
        F1, F2 = f1.transpose(1, 2), f2.transpose(1, 2)   # [B,HW,64]
        extra = orth_loss_naive(F1, F2) if orth == "naive" else orth_loss_gram(F1, F2)
    a1 = torch.sigmoid(f1 * d1) + d1                      # DBA.py:48-49
    a2 = torch.sigmoid(f2 * d2) + d2
    wf = p["conv_out_fg.weight"].reshape(EMB)
    wb = p["conv_out_bg.weight"].reshape(EMB)
    fg = torch.einsum("c,bcp->bp", wf, a1) + p["conv_out_fg.bias"]
    bg = torch.einsum("c,bcp->bp", wb, a2) + p["conv_out_bg.bias"]
    return fg.view(B, 1, H, W), bg.view(B, 1, H, W), extra


def rev_decoder_backward(x, p, gfg, gbg, gextra, with_dx=False):
    """Closed-form backward used by the HIP kernels (validated against autograd in tests).

    gfg,gbg: [B,1,H,W] upstream grads of the logits; gextra: python float, upstream grad of
    extra_loss.  Returns dict of parameter grads (same keys as ``p``) -- the input feature
    gradient is not needed (frozen backbone, loop_UCOD_DPL.py:148-158 feeds cached features).
    """
    B, C, H, W = x.shape
    HW = H * W
    X = x.reshape(B, C, HW)
    d1, d2, f1, f2, n1, n2 = _branches(x, p)
    e = p["learnable_embedding"]
    wf = p["conv_out_fg.weight"].reshape(EMB)
    wb = p["conv_out_bg.weight"].reshape(EMB)
    gf = gfg.reshape(B, 1, HW)
    gb = gbg.reshape(B, 1, HW)

    # orthogonality loss in Gram form (SURVEY.md section 7 hard parts)
    Z = B * HW * HW
    G1 = torch.bmm(f1, f1.transpose(1, 2))                # [B,64,64]
    G2 = torch.bmm(f2, f2.transpose(1, 2))
    s = (f1 * f2).sum(1, keepdim=True)                    # [B,1,HW]
    go1 = (2.0 * gextra / Z) * (torch.bmm(G2, f1) - s * f2)
    go2 = (2.0 * gextra / Z) * (torch.bmm(G1, f2) - s * f1)

    out = {}

    def branch(d, f, n, ev, w, g, go):
        sg = torch.sigmoid(f * d)
        a = sg + d
        dsg = sg * (1 - sg)
        ga = g * w.view(1, -1, 1)                         # dL/da
        gfeat = go + ga * dsg * d                         # dL/df (orth + gate path)
        # f = u / max(||u||, eps): projection backward over the pixel axis
        r = (f * gfeat).sum(2, keepdim=True)              # [B,64,1]
        clamped = (n <= NORM_EPS)
        gu = torch.where(clamped, gfeat / NORM_EPS, (gfeat - f * r) / n)
        gd = gu * ev.view(1, -1, 1) + ga * (dsg * f + 1.0)
        ge = (gu * d).sum((0, 2))                         # analytically ~0 (scale cancels)
        gw = (g * a).sum((0, 2))
        gbias = g.sum()
        return gd, ge, gw, gbias

    gd1, ge1, gwf, gbf = branch(d1, f1, n1, e[0], wf, gf, go1)
    gd2, ge2, gwb, gbb = branch(d2, f2, n2, e[1], wb, gb, go2)
    gd = torch.cat([gd1, gd2], 1)                         # [B,128,HW]
    out["decoupling.weight"] = torch.einsum("bnp,bcp->nc", gd, X).reshape(2 * EMB, C, 1, 1)
    out["decoupling.bias"] = gd.sum((0, 2))
    out["learnable_embedding"] = torch.stack([ge1, ge2])
    out["conv_out_fg.weight"] = gwf.reshape(1, EMB, 1, 1)
    out["conv_out_fg.bias"] = gbf.reshape(1)
    out["conv_out_bg.weight"] = gwb.reshape(1, EMB, 1, 1)
    out["conv_out_bg.bias"] = gbb.reshape(1)
    if with_dx:
        # gradient w.r.t. the input features (backbone-backward mode; pinned by g12_decoder_dx):
        # the 1x1 decoupling conv transposed, dX[b] = W^T gd[b]
        Wd = p["decoupling.weight"].reshape(2 * EMB, C)
        out["dx"] = torch.einsum("nc,bnp->bcp", Wd, gd).reshape(B, C, H, W)
    return out
