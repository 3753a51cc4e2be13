"""Oracle: DINOv2 / DINOv1 ViT backbone forward and the last-layer key hook.
TEST INFRASTRUCTURE ONLY.

The backbone arithmetic is NOT in /root/reference: data/utils/feature_extractor.py:20,25
instantiates HuggingFace ``transformers.AutoModel`` (requirement.txt:6, unpinned; 5.15.0
installed here).  ``dinov2_forward`` restates transformers==5.15.0
models/dinov2/modeling_dinov2.py (embeddings :38-149, eager attention :153-179, self-attention
:181-235, output :238-253, layer-scale :272-278, MLP :281-297, layer :342-381) on a flat
state dict with HF parameter names.  ``dinov1_forward`` restates the in-repo
models/backbones/dino.py:96-261 (fused qkv, no LayerScale, scale_factor+0.1 pos-embed
interpolation :202-221) with its parameter names.  Both return the tensor the reference's
hook captures (feature_extractor.py:42,46-47,55-58): the LAST layer's key projection
*including bias, before the head split*, CLS dropped, reshaped [B,C,h,w].
"""
import math
import torch
import torch.nn.functional as F

from .resize import torch_bicubic


def layer_norm(x, w, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _identity(t):
    return t


def autocast_rounding(dtype):
    """Rounding of a tensor to the autocast type and back (the f32 container keeps the arithmetic of this file unchanged).  Used to EMULATE what
    `accelerate launch --mixed_precision fp16` (scripts/launch_train_first_stage.sh:20) does to the reference's forward: torch's CUDA autocast runs
    linear / matmul / conv in the 16-bit type (operands rounded, f32 accumulation, result rounded) and layer_norm / softmax in f32; mixed adds
    promote to f32.  None = plain f32 (the cached-feature pass of the reference, base_dataset.py:135-138)."""
    return _identity if dtype is None else (lambda t: t.to(dtype).float())


def attention(q, k, v, heads, r=_identity):
    """eager_attention_forward (modeling_dinov2.py:153-179): softmax(QK^T * hd^-0.5) V.  ``r``: autocast rounding (see autocast_rounding)."""
    B, N, D = q.shape
    hd = D // heads
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    s = r(r(torch.matmul(r(q), r(k).transpose(2, 3))) * (hd ** -0.5))
    p = torch.softmax(s, dim=-1)
    return r(torch.matmul(r(p), r(v))).transpose(1, 2).reshape(B, N, D)


def patch_embed(img, w, b, patch):
    """conv(patch, stride patch) == per-patch dot product; flatten(2).transpose(1,2)."""
    return F.conv2d(img, w, b, stride=patch).flatten(2).transpose(1, 2)


def dinov2_pos_embed(pos, n_h, n_w):
    """modeling_dinov2.py:57-95.  pos [1,1+n0,D]."""
    n0 = pos.shape[1] - 1
    if n0 == n_h * n_w and n_h == n_w:
        return pos
    s = int(n0 ** 0.5)
    pp = pos[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2).float()
    pp = torch_bicubic(pp, n_h, n_w).to(pos.dtype)
    return torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, n_h * n_w, -1)), 1)


def lora_dropout_mask(seed, layer, proj, rows, D, p):
    """The counter-based LoRA-dropout mask of include/ucod_dpl.h (ucod_lora_dropout), restated with numpy uint32 arithmetic:
    [rows, D] float tensor of 0 or 1/(1-p)."""
    import numpy as np
    with np.errstate(over="ignore"):
        idx = np.arange(rows * D, dtype=np.uint64).astype(np.uint32)
        h = np.uint32(seed & 0xFFFFFFFF) ^ (idx * np.uint32(0x9E3779B1))
        h = h ^ (np.uint32((seed >> 32) & 0xFFFFFFFF) + np.uint32(layer) * np.uint32(0x85EBCA77))
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x7FEB352D)
        h = h ^ (h >> np.uint32(15))
        h = h * np.uint32(0x846CA68B)
        h = h ^ (h >> np.uint32(16))
    thresh = min(int(np.float32(p).astype(np.float64) * 1024.0), 1023)          # ABI 3: one mix per element, a 10-bit field per projection
    keep = ((h >> np.uint32(10 * proj)) & np.uint32(1023)) >= np.uint32(thresh)
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(thresh) / np.float32(1024.0))
    return torch.from_numpy(np.where(keep, inv, np.float32(0.0)).astype(np.float32).reshape(rows, D))


def _lora_linear(h, sd, name, lora_scale, mask=None, r=_identity):
    """nn.Linear, plus -- when the state dict carries ``<name>.lora_A.weight`` [r,D] / ``<name>.lora_B.weight`` [D,r] -- the
    peft LoRA branch the reference wraps query/key/value in (models/modules/full_model.py:47-72: r=2, lora_alpha=4,
    bias='none'; peft is not installed here, its published forward is ``base(x) + lora_B(lora_A(dropout(x))) * alpha/r``;
    dropout is the identity in this restatement -- SURVEY.md 8a row B9)."""
    y = r(r(h) @ r(sd[name + ".weight"]).t() + r(sd[name + ".bias"]))
    if name + ".lora_A.weight" in sd:
        hd = h if mask is None else h * mask.reshape(h.shape)       # nn.Dropout on lora_A's input (train mode)
        y = r(y + r(r(r(hd) @ r(sd[name + ".lora_A.weight"]).t()) @ r(sd[name + ".lora_B.weight"]).t()) * lora_scale)
    return y


def dinov2_forward(img, sd, heads, patch=14, eps=1e-6, n_layers=None, full_last_layer=True, lora_scale=2.0, lora_masks=None, autocast=None):
    """Returns (last_hidden_state [B,N,D] after the final LayerNorm, key [B,D,h,w]).  ``autocast`` = torch.float16 / torch.bfloat16: the same forward
    with the roundings torch's CUDA autocast would apply (autocast_rounding) -- the reference's launcher numerics as a second data point beside f32."""
    B, _, H, W = img.shape
    pre = "embeddings."
    r = autocast_rounding(autocast)
    x = r(patch_embed(r(img), r(sd[pre + "patch_embeddings.projection.weight"]), r(sd[pre + "patch_embeddings.projection.bias"]), patch))
    x = torch.cat((sd[pre + "cls_token"].expand(B, -1, -1), x), 1)
    x = x + dinov2_pos_embed(sd[pre + "position_embeddings"], H // patch, W // patch)
    L = n_layers if n_layers is not None else 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layer."))
    key = None
    gh, gw = H // patch, W // patch
    dinov2_forward.layer_keys = []                      # the key hook's map after EVERY layer (per-layer error budget of the device path)
    for i in range(L):
        p = f"encoder.layer.{i}."
        h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        a = p + "attention.attention."
        lm = (lambda nm: None) if lora_masks is None else (lambda nm: lora_masks.get((i, nm)))
        k = _lora_linear(h, sd, a + "key", lora_scale, lm("key"), r)
        dinov2_forward.layer_keys.append(k[:, 1:, :].reshape(B, gh, gw, -1).permute(0, 3, 1, 2))
        if i == L - 1:
            key = k
            dinov2_forward.last_ln1 = h                 # LN1 output of the last layer (CLS-attention row of the pseudo-label generator)
            if not full_last_layer:
                break
        q = _lora_linear(h, sd, a + "query", lora_scale, lm("query"), r)
        v = _lora_linear(h, sd, a + "value", lora_scale, lm("value"), r)
        o = attention(q, k, v, heads, r)
        o = r(r(o) @ r(sd[p + "attention.output.dense.weight"]).t() + r(sd[p + "attention.output.dense.bias"]))
        x = o * sd[p + "layer_scale1.lambda1"] + x
        h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = r(gelu_erf(r(r(h) @ r(sd[p + "mlp.fc1.weight"]).t() + r(sd[p + "mlp.fc1.bias"]))))
        h = r(h @ r(sd[p + "mlp.fc2.weight"]).t() + r(sd[p + "mlp.fc2.bias"]))
        x = h * sd[p + "layer_scale2.lambda1"] + x
    last = layer_norm(x, sd["layernorm.weight"], sd["layernorm.bias"], eps) if full_last_layer else None
    key_map = key[:, 1:, :].reshape(B, gh, gw, -1).permute(0, 3, 1, 2)   # feature_extractor.py:55-58
    return last, key_map


def dinov1_pos_embed(pos, n_h, n_w):
    """models/backbones/dino.py:202-221 (note: dino.py names them w0,h0 from (w,h)=x.shape[2:])."""
    n0 = pos.shape[1] - 1
    if n0 == n_h * n_w and n_h == n_w:
        return pos
    s = int(math.sqrt(n0))
    sf_h = (n_h + 0.1) / math.sqrt(n0)
    sf_w = (n_w + 0.1) / math.sqrt(n0)
    pp = pos[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2)
    oh, ow = int(math.floor(s * sf_h)), int(math.floor(s * sf_w))
    pp = torch_bicubic(pp, oh, ow, scale_h=sf_h, scale_w=sf_w)
    return torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, oh * ow, -1)), 1)


def dinov1_forward(img, sd, heads, patch=8, eps=1e-6, full_last_layer=True):
    """Returns (cls/norm output [B,N,D], key [B,D,h,w]) as ViTFeat(vit_feat='k') (dino.py:294-320)."""
    B, _, H, W = img.shape
    x = patch_embed(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], patch)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), 1)
    x = x + dinov1_pos_embed(sd["pos_embed"], H // patch, W // patch)
    L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    D = x.shape[-1]
    key = None
    for i in range(L):
        p = f"blocks.{i}."
        h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = h @ sd[p + "attn.qkv.weight"].t() + sd[p + "attn.qkv.bias"]
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        if i == L - 1:
            key = k
            if not full_last_layer:
                break
        o = attention(q, k, v, heads)
        x = x + (o @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"])
        h = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = gelu_erf(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        x = x + (h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"])
    last = layer_norm(x, sd["norm.weight"], sd["norm.bias"], eps) if full_last_layer else None
    gh, gw = H // patch, W // patch
    key_map = key[:, 1:, :].reshape(B, gh, gw, -1).permute(0, 3, 1, 2)
    return last, key_map


def dinov2_lora_grads(img, sd, heads, dkey, patch=14, eps=1e-6, lora_scale=2.0, lora_masks=None):
    """Backbone-backward mode: gradients of <key map, dkey> w.r.t. every LoRA matrix in ``sd``, by
    autograd over the restated forward.  ``dkey`` [B,D,h,w] is the cotangent arriving at the key hook
    (models/modules/full_model.py:95-106).  Returns (key [B,D,h,w], {param name: grad})."""
    names = [k for k in sd if ".lora_" in k]
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    sd2 = dict(sd)
    sd2.update(leaf)
    _, key = dinov2_forward(img, sd2, heads, patch=patch, eps=eps, full_last_layer=False, lora_scale=lora_scale, lora_masks=lora_masks)
    grads = torch.autograd.grad((key * dkey).sum(), [leaf[k] for k in names], allow_unused=True)
    return key.detach(), {k: (torch.zeros_like(leaf[k]) if g is None else g) for k, g in zip(names, grads)}
