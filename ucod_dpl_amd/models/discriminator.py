"""APM discriminator -- host-side mirror of models/discriminator.py (ConvBlock :15-70, Discriminator :73-95).

Same constructor (reads ``config.dis_use_features, config.dim, config.feature_size``), same state_dict names
(``maskConv.layers.{0,1}.*``, ``convs.{0,1}.layers.{0,1}.*``, ``linear.*``), all parameters start with
``requires_grad=False`` (:84-85), ``forward(mask, feature) -> [B,1]`` probabilities.  BatchNorm is applied with
batch statistics and the running buffers are updated on EVERY call, because the reference never switches this
module to eval (loop_UCOD_DPL.py:136).  The nn modules are parameter containers; the arithmetic is
``ucod_disc_fwd`` / ``ucod_disc_bwd``.  ``dis_use_features=True`` (:77-83,88-90: a dim->dim 3x3 ``featureConv`` whose output is concatenated
behind the mask branch; no shipped config enables it, configs/uscod/UCOD-DPL_dinov2.py:33) runs through generic kernels: forward
``ucod_unfold3x3`` + the exact-f32 MFMA GEMM ``ucod_dba_project`` + ``ucod_bn_lrelu_train_save`` + ``ucod_linear_sigmoid``; backward (the
discriminator phase) ``ucod_linear_sigmoid_bwd``, and per ConvBlock ``ucod_bn_lrelu_bwd`` + ``ucod_conv_wgrad_f32`` + (``ucod_dba_project`` with
the transposed weight + ``ucod_fold3x3`` for the input gradient).  Pinned on the real module's step (tests/golden G6b).
"""
import torch
from torch import nn

from .. import native as N, ops
from ..engine.registry import MODULE_REGISTRY

BN_EPS, BN_MOMENTUM, LRELU = 1e-5, 0.1, 0.1


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, leaky_relu_slope=0.1, bias=False, zero_init=False):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=bias),
                                    nn.BatchNorm2d(out_channels), nn.LeakyReLU(leaky_relu_slope, inplace=True))
        if zero_init:
            nn.init.constant_(self.layers[0].weight, 0)


class _DiscFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, module, *params):
        if not mask.is_cuda:
            raise RuntimeError("Discriminator runs on the HIP path only: move the module and its input to 'cuda'")
        mask = mask.float().contiguous()
        t = module.tensor_table()
        prob, saved = ops.disc_fwd(mask, t, update_running=True)
        module._bump_num_batches(t)
        ctx.module, ctx.saved, ctx.mask = module, saved, mask
        return prob.view(-1, 1)

    @staticmethod
    def backward(ctx, gprob):
        grads = ops.disc_bwd(ctx.mask, ctx.module.tensor_table(), ctx.saved, gprob.reshape(-1).float().contiguous())
        return (None, None) + tuple(grads)


class _DiscFeatFunction(torch.autograd.Function):
    """the feature-branch module under torch autograd (drop-in interface): forward / backward through the generic kernels"""

    @staticmethod
    def forward(ctx, mask, feature, module, *params):
        prob, saved = module.forward_features(mask, feature, save=True)
        ctx.module, ctx.saved = module, saved
        return prob.view(-1, 1)

    @staticmethod
    def backward(ctx, gprob):
        grads = [torch.empty_like(p) for p in ctx.module._param_list()]
        ctx.module.backward_features(ctx.saved, gprob.reshape(-1), grads, accumulate=False)
        return (None, None, None) + tuple(grads)


@MODULE_REGISTRY.register()
class Discriminator(nn.Module):
    FIELDS = ("w1", "g1", "b1", "w2", "g2", "b2", "w3", "g3", "b3", "lin_w", "lin_b")

    def __init__(self, config):
        super().__init__()
        self.maskConv = ConvBlock(1, 32, 3, 1, 1)
        self.use_features = config.dis_use_features
        if self.use_features:
            self.featureConv = ConvBlock(config.dim, config.dim, 3, 1, 1)
        indim = self.use_features * config.dim + 32
        outdim = indim // 2
        self.convs = nn.ModuleList([ConvBlock(indim // (2 ** i), outdim // (2 ** i), kernel_size=3, stride=2, padding=1) for i in range(2)])
        self.linear = nn.Linear(outdim // 2 * ((config.feature_size + 3) // 4) ** 2, 1)
        for p in self.parameters():
            p.requires_grad = False

    def _blocks(self):
        """the ConvBlocks in the reference's parameter order (named_parameters: maskConv, [featureConv], convs.0, convs.1)"""
        return (self.maskConv,) + ((self.featureConv,) if self.use_features else ()) + (self.convs[0], self.convs[1])

    def _param_list(self):
        out = []
        for b in self._blocks():
            out += [b.layers[0].weight, b.layers[1].weight, b.layers[1].bias]
        return out + [self.linear.weight, self.linear.bias]

    def tensor_table(self):
        blocks = (self.maskConv, self.convs[0], self.convs[1])
        t = dict(zip(self.FIELDS, (p.detach() for p in self._param_list())))
        for i, b in enumerate(blocks, 1):
            t[f"rm{i}"], t[f"rv{i}"] = b.layers[1].running_mean, b.layers[1].running_var
        # the three counters as one int64[3] (DiscArena makes them views of one tensor): ucod_disc_fwd bumps them in its last launch
        nbt = getattr(self, "_nbt", None)
        if nbt is not None and nbt.is_cuda and all(b.layers[1].num_batches_tracked.data_ptr() == nbt[i].data_ptr() for i, b in enumerate(blocks)):
            t["nbt"] = nbt
        return t

    def _bump_num_batches(self, table=None):
        """+1 on every block's num_batches_tracked -- unless the kernel call already did it (``table`` carries the counters: tensor_table())"""
        if table is not None and table.get("nbt") is not None:
            return
        nbt = getattr(self, "_nbt", None)           # set by DiscArena: the counters as views of one tensor
        if nbt is not None and all(b.layers[1].num_batches_tracked.data_ptr() == nbt[i].data_ptr() for i, b in enumerate(self._blocks())):
            nbt += 1
            return
        for b in self._blocks():
            b.layers[1].num_batches_tracked += 1

    # ---- dis_use_features=True: the generic kernels (csrc/disc_features.hip) -------------------------------------------------------------
    def _conv_block(self, x, block, stride, save):
        """ConvBlock.forward (:60-70) in training mode: conv3x3(pad 1, no bias) as unfold + exact-f32 GEMM, BatchNorm2d with batch statistics
        (running buffers updated), LeakyReLU(0.1).  ``save``: keep what the backward needs (the unfold, the conv output, the statistics)."""
        lib = N.load()
        conv, bn = block.layers[0], block.layers[1]
        B, Cin, H, W = x.shape
        Cout = conv.weight.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        K = Cin * 9
        Kpad = (K + 15) // 16 * 16
        cols = torch.empty(B, Kpad, Ho * Wo, dtype=torch.float32, device=x.device)
        N.check(lib.ucod_unfold3x3(N.ptr(x), N.ptr(cols), B, Cin, H, W, stride, Kpad, N.stream()), "ucod_unfold3x3")
        wmat = torch.zeros(Cout, Kpad, dtype=torch.float32, device=x.device)
        wmat[:, :K] = conv.weight.detach().reshape(Cout, K)
        y_pre = torch.empty(B, Cout, Ho * Wo, dtype=torch.float32, device=x.device)
        zero_b = torch.zeros(Cout, dtype=torch.float32, device=x.device)
        N.check(lib.ucod_dba_project(N.ptr(cols), N.ptr(wmat), N.ptr(zero_b), N.ptr(y_pre), B, Kpad, Ho * Wo, Cout, N.stream()), "ucod_dba_project")
        stats = torch.empty(lib.ucod_bn_lrelu_workspace_bytes(Cout), dtype=torch.uint8, device=x.device)
        g, b = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
        y = torch.empty_like(y_pre) if save else y_pre
        N.check(lib.ucod_bn_lrelu_train_save(N.ptr(y_pre), N.ptr(y), N.ptr(g), N.ptr(b), N.ptr(bn.running_mean), N.ptr(bn.running_var), B, Cout, Ho * Wo,
                                             BN_EPS, BN_MOMENTUM, LRELU, 1, N.ptr(stats), stats.numel(), N.stream()), "ucod_bn_lrelu_train_save")
        rec = dict(cols=cols, y_pre=y_pre, stats=stats, wmat=wmat, g=g, b=b, geom=(B, Cin, H, W, Cout, Ho, Wo, stride, K, Kpad)) if save else None
        return y.view(B, Cout, Ho, Wo), rec

    def forward_features(self, mask, feature, save=False):
        """-> (prob [B], saved): the feature-branch forward on device tensors; ``saved`` feeds ``backward_features``"""
        if feature is None:
            raise ValueError("Discriminator(dis_use_features=True).forward needs the feature map")
        if not mask.is_cuda:
            raise RuntimeError("Discriminator runs on the HIP path only: move the module and its inputs to 'cuda'")
        recs = {}
        with torch.no_grad():
            h, recs["mask"] = self._conv_block(mask.float().contiguous(), self.maskConv, 1, save)
            f, recs["feat"] = self._conv_block(feature.float().contiguous(), self.featureConv, 1, save)
            h = torch.cat((h, f), 1).contiguous()                     # (memory movement only)
            h, recs["c0"] = self._conv_block(h, self.convs[0], 2, save)
            h, recs["c1"] = self._conv_block(h, self.convs[1], 2, save)
            B = h.shape[0]
            x = h.reshape(B, -1).contiguous()
            out = torch.empty(B, dtype=torch.float32, device=h.device)
            lw, lb = self.linear.weight.detach().reshape(-1).contiguous(), self.linear.bias.detach().contiguous()
            N.check(N.load().ucod_linear_sigmoid(N.ptr(x), N.ptr(lw), N.ptr(lb), N.ptr(out), B, x.shape[1], N.stream()), "ucod_linear_sigmoid")
        self._bump_num_batches()
        recs.update(x=x, prob=out, lw=lw) if save else None
        return out, (recs if save else None)

    def _conv_block_bwd(self, rec, gout, g_w, g_gamma, g_beta, accumulate, need_input_grad):
        """gout [B,Cout,Ho*Wo] = dL/d(block output) -> gradients of the conv weight / BN affine (written or accumulated) and, when asked, dL/d(input)"""
        lib = N.load()
        B, Cin, H, W, Cout, Ho, Wo, stride, K, Kpad = rec["geom"]
        gy = torch.empty_like(rec["y_pre"])
        ws = torch.empty(lib.ucod_bn_lrelu_workspace_bytes(Cout), dtype=torch.uint8, device=gy.device)
        N.check(lib.ucod_bn_lrelu_bwd(N.ptr(rec["y_pre"]), N.ptr(gout), N.ptr(gy), N.ptr(rec["stats"]), N.ptr(rec["g"]), N.ptr(rec["b"]), N.ptr(g_gamma),
                                      N.ptr(g_beta), B, Cout, Ho * Wo, BN_EPS, LRELU, int(accumulate), N.ptr(ws), ws.numel(), N.stream()), "ucod_bn_lrelu_bwd")
        gwm = torch.empty(Cout, Kpad, dtype=torch.float32, device=gy.device)
        N.check(lib.ucod_conv_wgrad_f32(N.ptr(gy), N.ptr(rec["cols"]), N.ptr(gwm), B, Kpad, Ho * Wo, Cout, 0, N.stream()), "ucod_conv_wgrad_f32")
        gw_new = gwm[:, :K].reshape(g_w.shape)
        if accumulate:
            g_w.add_(gw_new)                                      # (two discriminator calls per step: the second adds)
        else:
            g_w.copy_(gw_new)
        if not need_input_grad:
            return None
        # dL/d(cols) = W^T gy (the same GEMM with the transposed weight; its K = Cout padded to the GEMM's 16), then the unfold's adjoint
        Cp = (Cout + 15) // 16 * 16
        wt = torch.zeros(Kpad, Cp, dtype=torch.float32, device=gy.device)
        wt[:, :Cout] = rec["wmat"].t()
        if Cp != Cout:
            gyp = torch.zeros(B, Cp, Ho * Wo, dtype=torch.float32, device=gy.device)
            gyp[:, :Cout] = gy
        else:
            gyp = gy
        gcols = torch.empty(B, Kpad, Ho * Wo, dtype=torch.float32, device=gy.device)
        zero_b = torch.zeros(Kpad, dtype=torch.float32, device=gy.device)
        N.check(lib.ucod_dba_project(N.ptr(gyp), N.ptr(wt), N.ptr(zero_b), N.ptr(gcols), B, Cp, Ho * Wo, Kpad, N.stream()), "ucod_dba_project")
        gx = torch.empty(B, Cin, H, W, dtype=torch.float32, device=gy.device)
        N.check(lib.ucod_fold3x3(N.ptr(gcols), N.ptr(gx), B, Cin, H, W, stride, Kpad, N.stream()), "ucod_fold3x3")
        return gx

    def backward_features(self, saved, gprob, grads, accumulate=False):
        """gradient of sum_b gprob[b] * prob[b] with respect to the 14 parameter tensors, into ``grads`` (tensors shaped like ``_param_list()``,
        written, or added to when ``accumulate``) -- loop_UCOD_DPL.py:248 ``accelerator.backward(loss)`` for this module"""
        lib = N.load()
        gm = dict(zip(("mask", "feat", "c0", "c1"), (grads[0:3], grads[3:6], grads[6:9], grads[9:12])))
        g_lw, g_lb = grads[12], grads[13]
        x = saved["x"]
        B, K = x.shape
        gx = torch.empty_like(x)
        glw = g_lw.view(-1)
        N.check(lib.ucod_linear_sigmoid_bwd(N.ptr(x), N.ptr(saved["lw"]), N.ptr(saved["prob"]), N.ptr(gprob.reshape(-1).float().contiguous()), N.ptr(gx),
                                            N.ptr(glw), N.ptr(g_lb), B, K, int(accumulate), N.stream()), "ucod_linear_sigmoid_bwd")
        c1, c0 = saved["c1"], saved["c0"]
        g = gx.view(B, c1["geom"][4], -1)
        g = self._conv_block_bwd(c1, g.contiguous(), *gm["c1"], accumulate, True)
        g = self._conv_block_bwd(c0, g.reshape(B, c0["geom"][4], -1).contiguous(), *gm["c0"], accumulate, True)
        nm = saved["mask"]["geom"][4]                                  # channels of the mask branch in the concatenation
        hw = g.shape[2] * g.shape[3]
        self._conv_block_bwd(saved["mask"], g[:, :nm].reshape(B, nm, hw).contiguous(), *gm["mask"], accumulate, False)
        self._conv_block_bwd(saved["feat"], g[:, nm:].reshape(B, g.shape[1] - nm, hw).contiguous(), *gm["feat"], accumulate, False)

    def _forward_with_features(self, mask, feature):
        params = self._param_list()
        if any(p.requires_grad for p in params) and torch.is_grad_enabled():
            return _DiscFeatFunction.apply(mask, feature, self, *params)
        return self.forward_features(mask, feature, save=False)[0].view(-1, 1)

    def forward(self, mask, feature=None):
        if self.use_features:
            return self._forward_with_features(mask, feature)
        return _DiscFunction.apply(mask, self, *self._param_list())
