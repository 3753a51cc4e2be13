"""CORAL second-stage refiner -- host-side mirror of models/UDLR.py::SparseRefiner (eval and training-mode forward).

Same constructor / ``from_config`` (reads ``config.window_size``, ``config.threshold``), same sub-module tree and therefore
the same state-dict names as the reference (``HRE.CSF.attn.{norm_q,norm_kv,attn.in_proj_*,attn.out_proj,mlp.0,mlp.2,norm_mlp}``,
``HRE.CSF.depthwise_conv``, ``HRE.CSF.mask_dec``, ``GE.alpha``, ``GE.fuser.{0,2}``) and the same
``forward(input_features, h_inputs, preds, h_targets=None) -> (outputs, ex_loss, opt)`` with the reference's ``opt`` keys
(UDLR.py:77-86).  nn modules are parameter containers (built in the reference's order, so a seeded default init reproduces
the reference's); the arithmetic is HIP: window gather + NCHW->token transpose, LayerNorm, bf16 MFMA projections, the
head_dim-96 cross-attention kernel, fused depthwise-7x7 + mask head, window scatter, gated ensembling.

``ex_loss`` is 0 in eval mode exactly as ``cal_ex_loss`` returns (UDLR.py:52-55).  In training mode (``.train()``) the forward is the same
arithmetic (every dropout of the module is 0, there is no BatchNorm) and ``cal_ex_loss`` adds the IoU-weighted window loss of
UDLR.py:56-75 from ``h_targets`` (HIP: ``ucod_window_loss``), with ``opt['window_targets']`` as the reference sets it.  The loss VALUE is
what the reference defines; it ships no second-stage training loop (``LocalRefineTrainLoop: pass``, engine/runner/loop_CORAL.py:38-39), so
no backward of the refiner exists to mirror and the returned loss carries no autograd graph.
"""
import math

import numpy as np
import torch
from torch import nn

from .. import native as N, ops
from ..engine.registry import MODULE_REGISTRY

LN_EPS = 1e-5
HEADS = 8


class CrossAttentionBlock(nn.Module):
    def __init__(self, dim, num_heads=8, mlp_ratio=4.0):
        super().__init__()
        self.norm_q = nn.LayerNorm(dim)
        self.norm_kv = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=num_heads, dropout=0.0, batch_first=True)
        self.drop_path = nn.Identity()
        hidden = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim), nn.Dropout(0.0))
        self.norm_mlp = nn.LayerNorm(dim)


class CSF(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.attn = CrossAttentionBlock(dim=dim)
        self.depthwise_conv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.mask_dec = nn.Conv2d(dim, 1, kernel_size=1, padding=0)


class HRE(nn.Module):
    def __init__(self, window_size, dim=768):
        super().__init__()
        self.window_size = window_size
        self.CSF = CSF()


class GatedEnsembler(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.alpha = nn.Parameter(torch.tensor(0.5))
        self.fuser = nn.Sequential(nn.Conv2d(num_classes, 64, kernel_size=1), nn.ReLU(), nn.Conv2d(64, num_classes, kernel_size=1))


class EntropySelector(nn.Module):
    def __init__(self, threshold, window_size):
        super().__init__()
        self.threshold = threshold
        self.window_size = window_size


@MODULE_REGISTRY.register()
class SparseRefiner(nn.Module):
    def __init__(self, config, window_size: int, threshold: float, dim: int = 768):
        super().__init__()
        self.config = config
        self.selector = EntropySelector(threshold, window_size)
        self.HRE = HRE(window_size, dim)
        self.GE = GatedEnsembler(1)
        self.window_size = window_size
        self.threshold = threshold
        self._prepared = None

    @classmethod
    def from_config(cls, config):
        return cls(config, config.window_size, config.threshold)

    # ------------------------------------------------------------------ weight preparation (bf16 GEMM operands, tap-major dw)
    def _prepare(self, dev):
        if self._prepared is not None and self._prepared["dev"] == dev:
            return self._prepared
        blk = self.HRE.CSF.attn
        C = blk.norm_q.weight.shape[0]
        hd = C // HEADS
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()  # noqa: E731
        W, b = f32(blk.attn.in_proj_weight), f32(blk.attn.in_proj_bias)
        qscale = torch.full((C,), (hd ** -0.5) * math.log2(math.e), dtype=torch.float32, device=dev)
        p = dict(dev=dev, C=C,
                 wq=ops.cast_bf16(W[:C].contiguous()), bq=b[:C].contiguous(), qscale=qscale,
                 wkv=ops.cast_bf16(W[C:].contiguous()), bkv=b[C:].contiguous(),
                 wo=ops.cast_bf16(f32(blk.attn.out_proj.weight)), bo=f32(blk.attn.out_proj.bias),
                 w1=ops.cast_bf16(f32(blk.mlp[0].weight)), b1=f32(blk.mlp[0].bias),
                 w2=ops.cast_bf16(f32(blk.mlp[2].weight)), b2=f32(blk.mlp[2].bias),
                 ones=torch.ones(C, dtype=torch.float32, device=dev),
                 nq=(f32(blk.norm_q.weight), f32(blk.norm_q.bias)), nkv=(f32(blk.norm_kv.weight), f32(blk.norm_kv.bias)),
                 nm=(f32(blk.norm_mlp.weight), f32(blk.norm_mlp.bias)),
                 dwT=f32(self.HRE.CSF.depthwise_conv.weight).reshape(C, 49).t().contiguous(), dwb=f32(self.HRE.CSF.depthwise_conv.bias),
                 mw=f32(self.HRE.CSF.mask_dec.weight).reshape(C), mb=float(self.HRE.CSF.mask_dec.bias.detach().item()),
                 f0w=f32(self.GE.fuser[0].weight).reshape(64), f0b=f32(self.GE.fuser[0].bias), f2w=f32(self.GE.fuser[2].weight).reshape(64),
                 f2b=float(self.GE.fuser[2].bias.detach().item()))
        self._prepared = p
        return p

    # ------------------------------------------------------------------ CSF on the selected windows (CSF.py:38-43)
    def _csf(self, P, l_feats, h_inputs, l_idx, h_idx, H, W):
        lib = N.load()
        dev, C = P["dev"], P["C"]
        Nw, HW = int(h_idx.numel()), H * W
        st = N.stream()
        q_tok = torch.empty(Nw * HW, C, dtype=torch.float32, device=dev)          # residual `query`, token-major
        c_tok = torch.empty(Nw * HW, C, dtype=torch.float32, device=dev)
        N.check(lib.ucod_gather_tokens(N.ptr(h_inputs), N.ptr(h_idx), N.ptr(q_tok), Nw, C, HW, st), "ucod_gather_tokens")
        N.check(lib.ucod_gather_tokens(N.ptr(l_feats), N.ptr(l_idx), N.ptr(c_tok), Nw, C, HW, st), "ucod_gather_tokens")
        qn = ops.layernorm(q_tok, *P["nq"], LN_EPS)
        cn = ops.layernorm(c_tok, *P["nkv"], LN_EPS)
        M = Nw * HW
        q = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16(N.EPI_BIAS_BF16, qn, P["wq"], q, M, C, C, bias=P["bq"], scale=P["qscale"])      # softmax scale folded in
        kv = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16(N.EPI_BIAS_BF16, cn, P["wkv"], kv, M, 2 * C, C, bias=P["bkv"])
        att = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        N.check(lib.ucod_cross_attention96_fwd(N.ptr(q), C, N.ptr(kv), kv.data_ptr() + C * 2, 2 * C, N.ptr(att), Nw, HW, HW, HEADS, st),
                "ucod_cross_attention96_fwd")
        x = ops.linear_scale_resid(att, P["wo"], P["bo"], P["ones"], q_tok)        # query + out_proj(attn)
        hn = ops.layernorm(x, *P["nm"], LN_EPS)
        g = ops.linear_bf16(hn, P["w1"], P["b1"], gelu=True)
        x = ops.linear_scale_resid(g, P["w2"], P["b2"], P["ones"], x)              # x + mlp(norm_mlp(x)); stays token-major
        win = torch.empty(Nw, 1, H, W, dtype=torch.float32, device=dev)
        N.check(lib.ucod_dwconv7_maskdec(N.ptr(x), N.ptr(P["dwT"]), N.ptr(P["dwb"]), N.ptr(P["mw"]), P["mb"], N.ptr(win), Nw, H, W, C, st),
                "ucod_dwconv7_maskdec")
        return win

    def cal_ex_loss(self, opt, win_flat=None):
        """UDLR.py:52-75.  Eval mode / no selected window: 0 (python int, as the reference returns).  Training mode: a 0-d f32 tensor."""
        loss = 0
        if not self.training:
            return loss, opt
        window_preds, preds, h_targets = opt["window_preds"], opt["preds"], opt["h_targets"]
        n = int(window_preds.shape[0])
        if n == 0:                                                          # mask.sum() == 0
            return loss, opt
        if h_targets is None:
            raise ValueError("SparseRefiner in training mode needs h_targets [B*ws*ws, 1, h, w] (models/UDLR.py:62)")
        lib, dev, ws = N.load(), window_preds.device, self.window_size
        h, w = window_preds.shape[-2:]
        B = preds.shape[0]
        h_targets = h_targets.to(dev, torch.float32).contiguous()
        if tuple(h_targets.shape) not in ((B * ws * ws, 1, h, w), (B * ws * ws, h, w)):
            raise ValueError(f"h_targets shape {tuple(h_targets.shape)} != {(B * ws * ws, 1, h, w)}")
        sel = h_targets.view(B * ws * ws, 1, h, w).index_select(0, win_flat.long())
        opt["window_targets"] = sel                                        # UDLR.py:71
        logits = int(sel.max().item() > 1)                                  # binary_iou's "already a probability?" test (one host sync)
        l_up = ops.bilinear_resize(preds, h * ws, w * ws)
        part = torch.empty(n, dtype=torch.float32, device=dev)
        ious = torch.empty(n, dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(lib.ucod_window_loss(N.ptr(window_preds), N.ptr(h_targets), N.ptr(win_flat), N.ptr(l_up), logits, N.ptr(part), N.ptr(ious), N.ptr(out),
                                     n, B, h, w, ws, N.stream()), "ucod_window_loss")
        opt["window_ious"] = ious
        return out[0], opt

    def forward(self, input_features, h_inputs, preds, h_targets=None):
        if not input_features.is_cuda:
            raise RuntimeError("SparseRefiner runs on the HIP path only: move the module and its inputs to 'cuda'")
        lib = N.load()
        dev = input_features.device
        P = self._prepare(dev)
        ws = self.window_size
        input_features = input_features.float().contiguous()
        h_inputs = h_inputs.float().contiguous()
        preds = preds.float().contiguous()
        B, C, H, W = input_features.shape
        ph, pw = preds.shape[-2:]
        st = N.stream()
        # EntropySelector (ASR.py:41-51); the "already a probability?" heuristic is a data-dependent branch -> one host sync
        lo, hi = torch.aminmax(preds)
        use_sigmoid = 0 if (lo.item() >= 0 and hi.item() <= 1) else 1
        entropy = torch.empty_like(preds)
        scores = torch.empty(B, ws, ws, dtype=torch.float32, device=dev)
        N.check(lib.ucod_entropy_scores(N.ptr(preds), use_sigmoid, N.ptr(entropy), N.ptr(scores), B, ph, pw, ws, st), "ucod_entropy_scores")
        mask = (scores > self.threshold).view(B, 1, ws, ws)
        mask_h = mask.view(B, ws * ws).cpu().numpy()
        sel = np.argwhere(mask_h)                                             # rows (b, j) in (b-major, raster) order
        coords_np = np.stack([sel[:, 1] // ws, sel[:, 1] % ws], 1).astype(np.int64) if len(sel) else np.zeros((0, 2), np.int64)
        coords_list = torch.from_numpy(coords_np).to(dev)
        Nw = len(sel)
        hH, hW = h_inputs.shape[-2:]
        if Nw > 0:
            l_idx = torch.from_numpy(sel[:, 0].astype(np.int32)).to(dev)
            h_idx = torch.from_numpy((sel[:, 0] * ws * ws + sel[:, 1]).astype(np.int32)).to(dev)
            window_preds = self._csf(P, input_features, h_inputs, l_idx, h_idx, hH, hW)
            cd = coords_list.to(torch.int32).contiguous()
        else:
            l_idx = torch.zeros(0, dtype=torch.int32, device=dev)
            h_idx = torch.zeros(0, dtype=torch.int32, device=dev)
            window_preds = torch.zeros(0, 1, hH, hW, dtype=torch.float32, device=dev)
            cd = torch.zeros(0, 2, dtype=torch.int32, device=dev)
        h_preds = torch.empty(B, 1, ws * hH, ws * hW, dtype=torch.float32, device=dev)
        N.check(lib.ucod_window_scatter(N.ptr(window_preds) if Nw else None, N.ptr(cd) if Nw else None, N.ptr(l_idx) if Nw else None,
                                        N.ptr(h_preds), Nw, B, hH, hW, ws, st), "ucod_window_scatter")
        # GatedEnsembler (GE_pix_level.py:16-25)
        h2, w2 = h_preds.shape[-2:]
        l1 = ops.bilinear_resize(preds, h2, w2)
        outputs = torch.empty_like(h_preds)
        ge_w = torch.empty_like(h_preds)
        wsb = torch.empty(lib.ucod_gated_ensemble_workspace_bytes(B, h2, w2), dtype=torch.uint8, device=dev)
        N.check(lib.ucod_gated_ensemble(N.ptr(l1), N.ptr(h_preds), N.ptr(P["f0w"]), N.ptr(P["f0b"]), N.ptr(P["f2w"]), P["f2b"], N.ptr(outputs),
                                        N.ptr(ge_w), N.ptr(wsb), B, h2, w2, st), "ucod_gated_ensemble")
        opt = {"mask": mask, "entropy": entropy, "h_preds": h_preds, "window_preds": window_preds, "GE_w": ge_w, "preds": preds,
               "coords_list": coords_list, "h_targets": h_targets}
        ex_loss, opt = self.cal_ex_loss(opt, h_idx)
        return outputs, ex_loss, opt
