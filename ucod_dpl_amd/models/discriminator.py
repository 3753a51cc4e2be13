"""APM discriminator -- host-side mirror of models/discriminator.py (ConvBlock :15-70, Discriminator :73-95).

Same constructor (reads ``config.dis_use_features, config.dim, config.feature_size``), same state_dict names
(``maskConv.layers.{0,1}.*``, ``convs.{0,1}.layers.{0,1}.*``, ``linear.*``), all parameters start with
``requires_grad=False`` (:84-85), ``forward(mask, feature) -> [B,1]`` probabilities.  BatchNorm is applied with
batch statistics and the running buffers are updated on EVERY call, because the reference never switches this
module to eval (loop_UCOD_DPL.py:136).  The nn modules are parameter containers; the arithmetic is
``ucod_disc_fwd`` / ``ucod_disc_bwd``.  ``dis_use_features=True`` (:77-83,88-90: a dim->dim 3x3 ``featureConv`` whose output is concatenated
behind the mask branch; no shipped config enables it, configs/uscod/UCOD-DPL_dinov2.py:33) runs FORWARD-only -- what the APM merge needs from
the frozen discriminator -- through generic kernels (``ucod_unfold3x3`` + the exact-f32 MFMA GEMM ``ucod_dba_project`` + ``ucod_bn_lrelu_train``
+ ``ucod_linear_sigmoid``); training that variant's parameters (the discriminator phase) is not built and raises.
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
        module._bump_num_batches()
        ctx.module, ctx.saved, ctx.mask = module, saved, mask
        return prob.view(-1, 1)

    @staticmethod
    def backward(ctx, gprob):
        grads = ops.disc_bwd(ctx.mask, ctx.module.tensor_table(), ctx.saved, gprob.reshape(-1).float().contiguous())
        return (None, None) + tuple(grads)


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

    def _param_list(self):
        blocks = (self.maskConv, self.convs[0], self.convs[1])
        out = []
        for b in blocks:
            out += [b.layers[0].weight, b.layers[1].weight, b.layers[1].bias]
        return out + [self.linear.weight, self.linear.bias]

    def tensor_table(self):
        blocks = (self.maskConv, self.convs[0], self.convs[1])
        t = dict(zip(self.FIELDS, (p.detach() for p in self._param_list())))
        for i, b in enumerate(blocks, 1):
            t[f"rm{i}"], t[f"rv{i}"] = b.layers[1].running_mean, b.layers[1].running_var
        return t

    def _bump_num_batches(self):
        nbt = getattr(self, "_nbt", None)           # set by DiscArena: the three counters as views of one tensor
        if nbt is not None and all(b.layers[1].num_batches_tracked.data_ptr() == nbt[i].data_ptr()
                                   for i, b in enumerate((self.maskConv, self.convs[0], self.convs[1]))):
            nbt += 1
            return
        for b in (self.maskConv, self.convs[0], self.convs[1]):
            b.layers[1].num_batches_tracked += 1

    # ---- dis_use_features=True: forward through the generic kernels (csrc/disc_features.hip) ---------------------------------------
    def _conv_block(self, x, block, stride):
        """ConvBlock.forward (:60-70) in training mode: conv3x3(pad 1, no bias) as unfold + exact-f32 GEMM, BatchNorm2d with batch statistics
        (running buffers updated), LeakyReLU(0.1)."""
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
        y = torch.empty(B, Cout, Ho * Wo, dtype=torch.float32, device=x.device)
        zero_b = torch.zeros(Cout, dtype=torch.float32, device=x.device)
        N.check(lib.ucod_dba_project(N.ptr(cols), N.ptr(wmat), N.ptr(zero_b), N.ptr(y), B, Kpad, Ho * Wo, Cout, N.stream()), "ucod_dba_project")
        ws = torch.empty(lib.ucod_bn_lrelu_workspace_bytes(Cout), dtype=torch.uint8, device=x.device)
        g, b = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
        N.check(lib.ucod_bn_lrelu_train(N.ptr(y), N.ptr(g), N.ptr(b), N.ptr(bn.running_mean), N.ptr(bn.running_var), B, Cout, Ho * Wo, BN_EPS, BN_MOMENTUM,
                                        LRELU, 1, N.ptr(ws), ws.numel(), N.stream()), "ucod_bn_lrelu_train")
        bn.num_batches_tracked += 1
        return y.view(B, Cout, Ho, Wo)

    def _forward_with_features(self, mask, feature):
        if any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            raise NotImplementedError("dis_use_features=True: only the forward of the frozen discriminator (APM merge) is built; its training "
                                      "(discriminator phase) has no HIP backward -- no shipped config enables this branch")
        if feature is None:
            raise ValueError("Discriminator(dis_use_features=True).forward needs the feature map")
        if not mask.is_cuda:
            raise RuntimeError("Discriminator runs on the HIP path only: move the module and its inputs to 'cuda'")
        with torch.no_grad():
            h = self._conv_block(mask.float().contiguous(), self.maskConv, 1)
            f = self._conv_block(feature.float().contiguous(), self.featureConv, 1)
            h = torch.cat((h, f), 1).contiguous()                     # (memory movement only)
            for blk in self.convs:
                h = self._conv_block(h, blk, 2)
            B = h.shape[0]
            x = h.reshape(B, -1).contiguous()
            out = torch.empty(B, dtype=torch.float32, device=h.device)
            lw, lb = self.linear.weight.detach().reshape(-1).contiguous(), self.linear.bias.detach().contiguous()
            N.check(N.load().ucod_linear_sigmoid(N.ptr(x), N.ptr(lw), N.ptr(lb), N.ptr(out), B, x.shape[1], N.stream()), "ucod_linear_sigmoid")
        return out.view(-1, 1)

    def forward(self, mask, feature=None):
        if self.use_features:
            return self._forward_with_features(mask, feature)
        return _DiscFunction.apply(mask, self, *self._param_list())
