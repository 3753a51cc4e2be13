"""Dual-Branch Adversarial decoder -- host-side mirror of models/modules/DBA.py::RevDecoder.

Same constructor (``RevDecoder(cfg, ema=False)``, reads ``cfg.dim``), same parameter names and shapes
(``decoupling.{weight,bias}``, ``learnable_embedding``, ``conv_out_{fg,bg}.{weight,bias}`` -- the shipped
``weights/UCOD_DPL_*.safetensors`` load strictly), same ``forward(x, get_bg_mask=False)`` returns
(DBA.py:31-59).  The arithmetic is the HIP path: exact-f32 MFMA projection, pixel-axis column norms,
fused gate+heads, Gram-form orthogonality loss, closed-form backward.  nn.Conv2d instances are kept only
as parameter containers; their forward is never called.
"""
import torch
import torch.nn as nn

from ... import ops
from ...engine.registry import MODULE_REGISTRY

EMB = 64


def pack_heads(dec):
    hw = torch.cat((dec.conv_out_fg.weight.reshape(EMB), dec.conv_out_bg.weight.reshape(EMB)))
    hb = torch.cat((dec.conv_out_fg.bias.reshape(1), dec.conv_out_bg.bias.reshape(1)))
    return hw.contiguous(), hb.contiguous()


class _DBAFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, bias, emb, wf, bf, wb, bb, want_extra):
        if not x.is_cuda:
            raise RuntimeError("RevDecoder runs on the HIP path only: move the module and its input to 'cuda'")
        x = x.float().contiguous()
        B, C, H, Wd = x.shape
        emb_f = emb.detach().reshape(2 * EMB).contiguous()
        hw = torch.cat((wf.detach().reshape(EMB), wb.detach().reshape(EMB))).contiguous()
        hb = torch.cat((bf.detach().reshape(1), bb.detach().reshape(1))).contiguous()
        d = ops.dba_project(x, W.detach().reshape(2 * EMB, C).contiguous(), bias.detach().contiguous())
        norm = ops.dba_colnorm(d, 0, emb_f)
        fg, bg, sdiag = ops.dba_heads(d, 0, emb_f, norm, hw, hb, want_bg=True, want_sdiag=want_extra)
        if want_extra:
            extra, gram = ops.orth_gram(d, 0, emb_f, norm, sdiag)
            extra = extra.reshape(())
        else:
            extra, gram = torch.zeros((), device=x.device), None
        ctx.save_for_backward(x, d, norm, gram, emb_f, hw)
        ctx.want_extra = want_extra
        ctx.W_t = W.detach().reshape(2 * EMB, C).t().contiguous() if ctx.needs_input_grad[0] else None
        return fg.view(B, 1, H, Wd), bg.view(B, 1, H, Wd), extra

    @staticmethod
    def backward(ctx, gfg, gbg, gextra):
        x, d, norm, gram, emb_f, hw = ctx.saved_tensors
        B, C, H, Wd = x.shape
        dev = x.device
        gfg = torch.zeros(B, H * Wd, device=dev) if gfg is None else gfg.reshape(B, H * Wd).float().contiguous()
        gbg = torch.zeros(B, H * Wd, device=dev) if gbg is None else gbg.reshape(B, H * Wd).float().contiguous()
        if gram is None:
            gram = torch.zeros(B, 2, EMB, EMB, device=dev)
            ge = 0.0
        else:
            ge = float(gextra) if gextra is not None else 0.0      # one host sync; the fused TrainLoop path has none
        gd, ghw, ghb, gdb = ops.dba_bwd(d, 0, emb_f, norm, hw, gram, gfg, gbg, ge)
        gW = ops.dba_wgrad(gd, x)
        gx = None
        if ctx.needs_input_grad[0]:
            # backbone-backward mode (models/modules/full_model.py:108-126): the 1x1 conv transposed, dX[b] = W^T gd[b],
            # on the same exact-f32 MFMA kernel as the forward projection (input channels = the 128 decoder channels)
            Wt = ctx.W_t
            gx = ops.dba_project(gd.view(B, 2 * EMB, H, Wd), Wt, torch.zeros(C, device=dev)).view(B, C, H, Wd)
        return (gx, gW.view(2 * EMB, C, 1, 1), gdb, torch.zeros(2, EMB, device=dev), ghw[0].reshape(1, EMB, 1, 1), ghb[0:1].clone(),
                ghw[1].reshape(1, EMB, 1, 1), ghb[1:2].clone(), None)


@MODULE_REGISTRY.register()
class RevDecoder(nn.Module):
    def __init__(self, cfg, ema: bool = False):
        super().__init__()
        feature_dim = cfg.dim
        self.embed_dim = EMB
        self.decoupling = nn.Conv2d(feature_dim, 2 * EMB, kernel_size=(1, 1))
        self.learnable_embedding = nn.Parameter(torch.randn(2, EMB))
        self.conv_out_fg = nn.Conv2d(EMB, 1, kernel_size=(1, 1))
        self.conv_out_bg = nn.Conv2d(EMB, 1, kernel_size=(1, 1))
        self.ema = ema
        if ema:
            for p in self.parameters():
                p.detach_()                                   # DBA.py:21-23: the teacher never receives gradients

    def forward(self, x, get_bg_mask=False):
        if type(x) is list:
            x = x[-1]
        fg, bg, extra = _DBAFunction.apply(x, self.decoupling.weight, self.decoupling.bias, self.learnable_embedding,
                                           self.conv_out_fg.weight, self.conv_out_fg.bias, self.conv_out_bg.weight,
                                           self.conv_out_bg.bias, not self.ema)
        if not self.ema:
            return fg, bg, extra
        if get_bg_mask:
            return fg, bg
        return fg
