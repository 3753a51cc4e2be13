"""APM discriminator -- host-side mirror of models/discriminator.py (ConvBlock :15-70, Discriminator :73-95).

Same constructor (reads ``config.dis_use_features, config.dim, config.feature_size``), same state_dict names
(``maskConv.layers.{0,1}.*``, ``convs.{0,1}.layers.{0,1}.*``, ``linear.*``), all parameters start with
``requires_grad=False`` (:84-85), ``forward(mask, feature) -> [B,1]`` probabilities.  BatchNorm is applied with
batch statistics and the running buffers are updated on EVERY call, because the reference never switches this
module to eval (loop_UCOD_DPL.py:136).  The nn modules are parameter containers; the arithmetic is
``ucod_disc_fwd`` / ``ucod_disc_bwd``.  ``dis_use_features=True`` (a 768->768 3x3 conv no shipped config
enables, configs/uscod/UCOD-DPL_dinov2.py:33) is rejected.
"""
import torch
from torch import nn

from .. import ops
from ..engine.registry import MODULE_REGISTRY


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
            raise NotImplementedError("dis_use_features=True is not on the shipped configs' path (UCOD-DPL_dinov2.py:33)")
        self.convs = nn.ModuleList([ConvBlock(32 // (2 ** i), 16 // (2 ** i), kernel_size=3, stride=2, padding=1) for i in range(2)])
        self.linear = nn.Linear(8 * ((config.feature_size + 3) // 4) ** 2, 1)
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
        for b in (self.maskConv, self.convs[0], self.convs[1]):
            b.layers[1].num_batches_tracked += 1

    def forward(self, mask, feature=None):
        return _DiscFunction.apply(mask, self, *self._param_list())
