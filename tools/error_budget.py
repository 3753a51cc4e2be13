"""Per-op-class error budget of the backbone's 16-bit roundings on the TRAINED-LIKE weights (VERDICT r5 next #1c).

CPU only.  The f32 forward of oracle/vit.py::dinov2_forward is restated here with the DEVICE path's rounding points (where ucod_vit_forward rounds a
value to the 16-bit operand type or to the fp16 residual stream -- csrc/vit.hip), each rounding point belonging to one op class that can be switched
on or off:

    patch   im2col patches + patch weights (A / B operands of the patch-embedding GEMM)
    stream  the residual stream x held in IEEE fp16 (after the embeddings, after the out-projection update, after the fc2 update)
    qkv     LayerNorm-1 output h and W_qkv (operands of the QKV GEMM)
    qk      q (pre-scaled) and k as the attention kernel reads them (operands of Q K^T)
    pv      the probabilities p and v (operands of P V)
    proj    the attention output and W_proj (operands of the out-projection)
    fc1     LayerNorm-2 output and W_fc1
    fc2     GELU output and W_fc2
    key     last layer's LayerNorm-1 output and W_k (operands of the key hook)

For every class: the logit max-abs error against the all-f32 forward with ONLY that class rounded ("alone") and with every class BUT that one rounded
("all but").  The decoder runs in f32 (its device form is f32-equivalent: gemm_split.hip).  Output: one JSON object per line + a table.

    python tools/error_budget.py [--half f16|bf16] [--images 2] [--out profiles/r06_error_budget.json]
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decoder as OD, vit as OV  # noqa: E402
from oracle.resize import torch_bilinear  # noqa: E402
from ucod_dpl_amd.data.utils.feature_extractor import trained_like_state_dict, random_state_dict, ARCHS  # noqa: E402

CLASSES = ("patch", "stream", "qkv", "qk", "pv", "proj", "fc1", "fc2", "key")


def split_round(t, dtype, terms):
    """t as the sum of `terms` values of `dtype` (hi + lo [+ lo2]): what the split-operand kernels multiply by."""
    out = torch.zeros_like(t)
    rest = t
    for _ in range(terms):
        part = rest.to(dtype).float()
        out = out + part
        rest = rest - part
    return out


def forward(img, sd, heads, on, dtype, patch=14, eps=1e-6, terms=1):
    """Device-like forward: `on` = set of classes whose rounding points are active; terms > 1 = split operands (hi + lo ...)."""
    r = lambda c: ((lambda t: split_round(t, dtype, terms)) if c in on else (lambda t: t))  # noqa: E731
    rs = (lambda t: t.to(torch.float16).float()) if "stream" in on else (lambda t: t)
    B, _, H, W = img.shape
    pre = "embeddings."
    x = OV.patch_embed(r("patch")(img), r("patch")(sd[pre + "patch_embeddings.projection.weight"]), sd[pre + "patch_embeddings.projection.bias"], patch)
    x = torch.cat((sd[pre + "cls_token"].expand(B, -1, -1), x), 1)
    x = rs(x + OV.dinov2_pos_embed(sd[pre + "position_embeddings"], H // patch, W // patch))
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layer."))
    D = x.shape[-1]
    hd = D // heads
    gh, gw = H // patch, W // patch
    for i in range(L):
        p = f"encoder.layer.{i}."
        a = p + "attention.attention."
        h = OV.layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        if i == L - 1:
            k = r("key")(h) @ r("key")(sd[a + "key.weight"]).t() + sd[a + "key.bias"]
            return k[:, 1:, :].reshape(B, gh, gw, -1).permute(0, 3, 1, 2)
        hq = r("qkv")(h)
        q, k, v = (hq @ r("qkv")(sd[a + f"{n}.weight"]).t() + sd[a + f"{n}.bias"] for n in ("query", "key", "value"))
        q = r("qk")(q * (hd ** -0.5 * math.log2(math.e)))           # the QKV epilogue pre-scales Q before its rounding (csrc/vit.hip:86-92)
        k = r("qk")(k)
        v = r("pv")(v)
        sh = lambda t: t.view(B, -1, heads, hd).transpose(1, 2)  # noqa: E731
        s = torch.matmul(sh(q), sh(k).transpose(2, 3))
        e = torch.exp2(s - s.amax(-1, keepdim=True))
        o = torch.matmul(r("pv")(e), sh(v)) / e.sum(-1, keepdim=True)   # row sum of the unrounded exponentials, in f32
        o = o.transpose(1, 2).reshape(B, -1, D)
        o = r("proj")(o) @ r("proj")(sd[p + "attention.output.dense.weight"]).t() + sd[p + "attention.output.dense.bias"]
        x = rs(o * sd[p + "layer_scale1.lambda1"] + x)
        h = OV.layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = OV.gelu_erf(r("fc1")(h) @ r("fc1")(sd[p + "mlp.fc1.weight"]).t() + sd[p + "mlp.fc1.bias"])
        h = r("fc2")(h) @ r("fc2")(sd[p + "mlp.fc2.weight"]).t() + sd[p + "mlp.fc2.bias"]
        x = rs(h * sd[p + "layer_scale2.lambda1"] + x)
    raise AssertionError


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--half", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--weights", default="trained_like", choices=["trained_like", "flat"])
    ap.add_argument("--terms", type=int, default=1, help="operands as a sum of this many 16-bit values (split-operand kernels)")
    ap.add_argument("--sets", default="", help="extra class sets to evaluate, e.g. 'qk+pv,qkv+qk' (rounded classes)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dtype = torch.float16 if a.half == "f16" else torch.bfloat16
    arch = "dinov2_vitb14"
    D, heads, L, P, _, _ = ARCHS[arch]
    sd = (trained_like_state_dict if a.weights == "trained_like" else random_state_dict)(arch, 0, 518)
    img = torch.randn(a.images, 3, 518, 518, generator=torch.Generator().manual_seed(2024))
    dec = OD.init_params(D, torch.Generator().manual_seed(42))

    def logits(key):
        return OD.rev_decoder_forward(torch_bilinear(key, 68, 68), dec, orth="gram")[0]

    rows = []
    with torch.no_grad():
        t0 = time.time()
        key0 = forward(img, sd, heads, set(), dtype)
        _, key_ref = OV.dinov2_forward(img, sd, heads=heads, patch=P, eps=1e-6, full_last_layer=False)
        fg0 = logits(key_ref)
        base = float((logits(key0) - fg0).abs().max())
        print(f"# restated forward vs oracle (no rounding): logit max-abs {base:.2e}; one forward {time.time() - t0:.0f} s", flush=True)

        def run(name, on, terms=a.terms):
            key = forward(img, sd, heads, set(on), dtype, terms=terms)
            fg = logits(key)
            row = dict(case=name, rounded=sorted(on), half=a.half, terms=terms, weights=a.weights, logit_max_abs=float((fg - fg0).abs().max()),
                       key_rel_l2=float(((key - key_ref).double().norm() / key_ref.double().norm())),
                       mask_flipped_fraction=float(((fg > 0) != (fg0 > 0)).float().mean()), ref_logit_abs_max=float(fg0.abs().max()))
            rows.append(row)
            print(json.dumps(row), flush=True)

        run("all", CLASSES)
        for c in CLASSES:
            run(f"alone:{c}", (c,))
        for c in CLASSES:
            run(f"all_but:{c}", tuple(x for x in CLASSES if x != c))
        for s in filter(None, a.sets.split(",")):
            run(f"set:{s}", tuple(s.split("+")))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(tool="tools/error_budget.py", images=a.images, arch=arch, size=518, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
