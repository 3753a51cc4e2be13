"""Oracle: APM discriminator.  TEST INFRASTRUCTURE ONLY.

Restates models/discriminator.py:60-70 (ConvBlock = conv3x3 no-bias + BatchNorm2d +
LeakyReLU(0.1)) and :86-95 (Discriminator.forward) for the shipped configuration
``dis_use_features=False`` (configs/uscod/UCOD-DPL_dinov2.py:33).  BatchNorm is always in
training mode in the reference (nobody calls .eval() on it, loop_UCOD_DPL.py:136), so each
call normalises with its own batch statistics and mutates the running buffers.

State dict keys (reference names):
  maskConv.layers.0.weight [32,1,3,3]; maskConv.layers.1.{weight,bias,running_mean,running_var,num_batches_tracked}
  convs.{0,1}.layers.0.weight [16,32,3,3]/[8,16,3,3]; convs.{0,1}.layers.1.*
  linear.weight [1, 8*ceil(fs/4)^2]; linear.bias [1]
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU = 0.1
BLOCKS = (("maskConv", 1, 32, 1), ("convs.0", 32, 16, 2), ("convs.1", 16, 8, 2))


def init_state(feature_size, generator=None, dtype=torch.float32):
    """nn.Conv2d / nn.BatchNorm2d / nn.Linear default initialisation."""
    g = generator
    sd = {}
    for name, cin, cout, _ in BLOCKS:
        bound = 1.0 / math.sqrt(cin * 9)
        sd[f"{name}.layers.0.weight"] = (torch.rand(cout, cin, 3, 3, generator=g, dtype=dtype) * 2 - 1) * bound
        sd[f"{name}.layers.1.weight"] = torch.ones(cout, dtype=dtype)
        sd[f"{name}.layers.1.bias"] = torch.zeros(cout, dtype=dtype)
        sd[f"{name}.layers.1.running_mean"] = torch.zeros(cout, dtype=dtype)
        sd[f"{name}.layers.1.running_var"] = torch.ones(cout, dtype=dtype)
        sd[f"{name}.layers.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    nin = 8 * ((feature_size + 3) // 4) ** 2              # discriminator.py:83
    bound = 1.0 / math.sqrt(nin)
    sd["linear.weight"] = (torch.rand(1, nin, generator=g, dtype=dtype) * 2 - 1) * bound
    sd["linear.bias"] = (torch.rand(1, generator=g, dtype=dtype) * 2 - 1) * bound
    return sd


def conv_block(x, sd, name, stride, update_running=True):
    """discriminator.py:42-46,60-70 with train-mode batch statistics."""
    y = F.conv2d(x, sd[f"{name}.layers.0.weight"], None, stride=stride, padding=1)
    n = y.numel() // y.shape[1]
    mean = y.mean((0, 2, 3))
    var = ((y - mean.view(1, -1, 1, 1)) ** 2).mean((0, 2, 3))            # biased, used to normalise
    if update_running:
        with torch.no_grad():
            unb = var * (n / max(n - 1, 1))
            rm, rv = f"{name}.layers.1.running_mean", f"{name}.layers.1.running_var"
            sd[rm] = (1 - BN_MOMENTUM) * sd[rm] + BN_MOMENTUM * mean.detach()
            sd[rv] = (1 - BN_MOMENTUM) * sd[rv] + BN_MOMENTUM * unb.detach()
            sd[f"{name}.layers.1.num_batches_tracked"] = sd[f"{name}.layers.1.num_batches_tracked"] + 1
    yh = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    yh = yh * sd[f"{name}.layers.1.weight"].view(1, -1, 1, 1) + sd[f"{name}.layers.1.bias"].view(1, -1, 1, 1)
    return torch.where(yh >= 0, yh, yh * LRELU)


def discriminator_forward(mask, sd, update_running=True):
    """discriminator.py:86-95.  mask [B,1,H,W] -> prob [B,1].  ``sd`` is updated in place
    (running statistics) exactly as the always-train-mode reference module does."""
    h = mask
    for name, _, _, stride in BLOCKS:
        h = conv_block(h, sd, name, stride, update_running)
    h = h.flatten(1)
    z = h @ sd["linear.weight"].t() + sd["linear.bias"]
    return torch.sigmoid(z)


def discriminator_forward_with_features(mask, feature, sd, update_running=True):
    """discriminator.py:86-95 with ``dis_use_features=True`` (:77-83): featureConv (dim -> dim) on the feature map, concatenated behind the
    mask branch, two stride-2 ConvBlocks on (dim + 32) and (dim + 32) // 2 channels, Linear on (dim + 32) // 4 * ((fs + 3) // 4)**2."""
    h = conv_block(mask, sd, "maskConv", 1, update_running)
    f = conv_block(feature, sd, "featureConv", 1, update_running)
    h = torch.cat((h, f), 1)
    h = conv_block(h, sd, "convs.0", 2, update_running)
    h = conv_block(h, sd, "convs.1", 2, update_running)
    z = h.flatten(1) @ sd["linear.weight"].t() + sd["linear.bias"]
    return torch.sigmoid(z)
