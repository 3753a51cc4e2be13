"""GPU parity tests of the backbone-backward mode, every call through the C ABI.

Reference = oracle/vit.py + torch autograd on the CPU (pinned against HuggingFace Dinov2Model autograd by
tests/golden/g12_lora_backbone.npz; the reference's own full_model.py is not importable).  bf16 kernels are compared
with an f32/f64 evaluation of the same bf16-rounded operands; the end-to-end LoRA gradients carry the accumulated
bf16 error of a full forward + backward and are held to a relative-L2 bar instead (stated per test).
"""
import ctypes as C
import math

import pytest
import torch

from conftest import load_golden, sub, maxdiff

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from ucod_dpl_amd import native as N, ops  # noqa: E402
from ucod_dpl_amd.vit_engine import ViTLoRAEngine  # noqa: E402
from oracle import vit as OV  # noqa: E402

DEV = "cuda"
AUG = N.LORA_AUG


def bf(t):
    return t.to(torch.bfloat16)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def lora_arena(r, D, g, b_scale=0.05):
    """one layer: [A_q | B_q | A_k | B_k | A_v | B_v]"""
    parts, mats = [], []
    for _ in range(3):
        A = torch.randn(r, D, generator=g) / math.sqrt(D)
        Bm = torch.randn(D, r, generator=g) * b_scale
        parts += [A.reshape(-1), Bm.reshape(-1)]
        mats.append((A, Bm))
    return torch.cat(parts), mats


# ------------------------------------------------------------------------------------------------ row-wise kernels
@pytest.mark.parametrize("M,D,r", [(37, 128, 2), (300, 768, 2), (64, 1024, 4), (50, 384, 1)])
def test_layernorm_lora(M, D, r):
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 2 + 0.3
    gam, bet = torch.randn(D, generator=g), torch.randn(D, generator=g)
    flat, mats = lora_arena(r, D, g)
    out = torch.full((M, D + AUG), 7.0, dtype=torch.bfloat16, device=DEV)
    xs, gs, bs, fs = x.to(DEV), gam.to(DEV), bet.to(DEV), flat.to(DEV)
    N.check(N.load().ucod_layernorm_lora(N.ptr(xs), N.ptr(gs), N.ptr(bs), N.ptr(fs), r, N.ptr(out), M, D, 1e-6, None, N.stream()), "ln_lora")
    out = out.float().cpu()
    h = OV.layer_norm(x.double(), gam.double(), bet.double(), 1e-6)
    assert maxdiff(out[:, :D], h) < 2e-2 * max(1.0, h.abs().max().item())
    u = torch.cat([h @ A.double().t() for A, _ in mats], 1)
    assert maxdiff(out[:, D:D + 3 * r], u) < 1e-2 * max(1.0, u.abs().max().item())
    assert float(out[:, D + 3 * r:].abs().max()) == 0.0


@pytest.mark.parametrize("M,D", [(37, 128), (300, 768), (20, 1024)])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_bwd(M, D, with_res):
    g = torch.Generator().manual_seed(M * 3 + D)
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).requires_grad_(True)
    gam = torch.randn(D, generator=g)
    dy = torch.randn(M, D, generator=g)
    dres = torch.randn(M, D, generator=g) if with_res else None
    sc = torch.rand(D, generator=g) + 0.5
    y = OV.layer_norm(x.double(), gam.double(), torch.zeros(D).double(), 1e-6)
    (gx,) = torch.autograd.grad((y * dy.double()).sum(), x)
    ref = gx + (dres.double() if with_res else 0)
    dx = torch.empty(M, D, device=DEV)
    s = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    dys, xs, gs, scs = dy.to(DEV), x.detach().to(DEV), gam.to(DEV), sc.to(DEV)
    drs = dres.to(DEV) if with_res else None
    N.check(N.load().ucod_layernorm_bwd(N.ptr(dys), N.ptr(xs), N.ptr(gs), N.ptr(drs) if with_res else None, N.ptr(scs), N.ptr(dx), N.ptr(s), M, D,
                                        1e-6, N.stream()), "ln_bwd")
    assert maxdiff(dx.cpu(), ref) < 2e-5 * max(1.0, ref.abs().max().item())
    assert maxdiff(s.float().cpu(), ref * sc.double()) < 1e-2 * max(1.0, (ref * sc.double()).abs().max().item())


@pytest.mark.parametrize("M,D", [(37, 128), (300, 768), (20, 1024)])
def test_layernorm_bwd_with_16_bit_inputs_equals_the_f32_form_on_rounded_input(M, D):
    """ucod_layernorm_bwd_ex: dy as bf16 (what ucod_vit_backward feeds from its dgrad GEMMs) and x as fp16 (the saved residual stream of a training pass with
    vit.resid16) == ucod_layernorm_bwd on the same values held in f32."""
    g = torch.Generator().manual_seed(M * 5 + D)
    x16 = (torch.randn(M, D, generator=g) * 2 + 0.3).to(torch.float16).to(DEV)
    x32 = x16.float()
    gam, sc = torch.randn(D, generator=g).to(DEV), (torch.rand(D, generator=g) + 0.5).to(DEV)
    dy16 = torch.randn(M, D, generator=g).to(torch.bfloat16).to(DEV)
    dy32 = dy16.float()
    dres = torch.randn(M, D, generator=g).to(DEV)
    lib, out = N.load(), {}
    for name, dy, x, flags in (("f32", dy32, x32, None), ("dy16", dy16, x32, 1), ("both16", dy16, x16, 3)):
        dx = torch.empty(M, D, device=DEV)
        s = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        if flags is None:
            N.check(lib.ucod_layernorm_bwd(N.ptr(dy), N.ptr(x), N.ptr(gam), N.ptr(dres), N.ptr(sc), N.ptr(dx), N.ptr(s), M, D, 1e-6, N.stream()), name)
        else:
            N.check(lib.ucod_layernorm_bwd_ex(N.ptr(dy), N.ptr(x), flags, N.ptr(gam), N.ptr(dres), N.ptr(sc), N.ptr(dx), N.ptr(s), M, D, 1e-6, N.stream()), name)
        out[name] = (dx.cpu(), s.float().cpu())
    for name in ("dy16", "both16"):
        assert torch.equal(out["f32"][0], out[name][0]) and torch.equal(out["f32"][1], out[name][1]), name
    assert lib.ucod_layernorm_bwd_ex(N.ptr(dy32), N.ptr(x16), 2, N.ptr(gam), N.ptr(dres), N.ptr(sc), N.ptr(dx), N.ptr(s), M, D, 1e-6, N.stream()) != 0   # fp16 x only with bf16 dy


def test_key_grad_tokens_and_lora_pack():
    B, tok, D, r = 2, 26, 128, 2
    g = torch.Generator().manual_seed(5)
    dkey = torch.randn(B, D, tok - 1, generator=g)
    out = torch.full((B * tok, 3 * D + AUG), 3.0, dtype=torch.bfloat16, device=DEV)
    dks = dkey.to(DEV)
    N.check(N.load().ucod_key_grad_tokens(N.ptr(dks), N.ptr(out), B, tok, D, N.stream()), "key_grad")
    o = out.float().cpu().reshape(B, tok, 3 * D + AUG)
    assert float(o[:, :, :D].abs().max()) == 0 and float(o[:, :, 2 * D:].abs().max()) == 0 and float(o[:, 0].abs().max()) == 0
    assert maxdiff(o[:, 1:, D:2 * D], bf(dkey.transpose(1, 2)).float()) == 0
    flat, mats = lora_arena(r, D, g)
    w = torch.zeros(3 * D, D + AUG, dtype=torch.bfloat16, device=DEV)
    wt = torch.zeros(D, 3 * D + AUG, dtype=torch.bfloat16, device=DEV)
    fs = flat.to(DEV)
    N.check(N.load().ucod_lora_pack(N.ptr(fs), r, 2.0, N.ptr(w), N.ptr(wt), D, 0, N.stream()), "lora_pack")
    w, wt = w.float().cpu(), wt.float().cpu()
    for p, (A, Bm) in enumerate(mats):
        blk = w[p * D:(p + 1) * D, D:]
        assert maxdiff(blk[:, p * r:(p + 1) * r], bf(2.0 * Bm).float()) == 0
        blk = blk.clone()
        blk[:, p * r:(p + 1) * r] = 0
        assert float(blk.abs().max()) == 0
        assert maxdiff(wt[:, 3 * D + p * r:3 * D + (p + 1) * r], bf(A.t()).float()) == 0
    assert float(wt[:, 3 * D + 3 * r:].abs().max()) == 0


@pytest.mark.parametrize("M,D,r", [(52, 128, 2), (3000, 768, 2), (500, 256, 3), (700, 384, 1)])
def test_lora_grad(M, D, r):
    g = torch.Generator().manual_seed(M + r)
    scaling = 2.0
    flat, mats = lora_arena(r, D, g)
    dqkv = bf(torch.randn(M, 3 * D, generator=g))
    h = bf(torch.randn(M, D, generator=g))
    u = bf(torch.cat([h.float() @ A.t() for A, _ in mats], 1))
    d_aug = torch.zeros(M, 3 * D + AUG, dtype=torch.bfloat16)
    d_aug[:, :3 * D] = dqkv
    h_aug = torch.zeros(M, D + AUG, dtype=torch.bfloat16)
    h_aug[:, :D] = h
    h_aug[:, D:D + 3 * r] = u
    lib = N.load()
    wsb = lib.ucod_lora_grad_workspace_bytes(D)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    grad = torch.full((6 * r * D,), 9.0, device=DEV)
    ds, hs, fs = d_aug.to(DEV), h_aug.to(DEV), flat.to(DEV)
    N.check(lib.ucod_lora_grad(N.ptr(ds), N.ptr(hs), N.ptr(fs), r, scaling, N.ptr(grad), 0, N.ptr(ws), wsb, M, D, None, N.stream()), "lora_grad")
    grad, t_out = grad.cpu(), ds.float().cpu()[:, 3 * D:]
    off = 0
    for p, (A, Bm) in enumerate(mats):
        dq = dqkv[:, p * D:(p + 1) * D].double()
        t = scaling * dq @ Bm.double()
        assert maxdiff(t_out[:, p * r:(p + 1) * r], t) < 1e-2 * max(1.0, t.abs().max().item())
        tb = t_out[:, p * r:(p + 1) * r].double()                        # the kernel uses the bf16-rounded t for dA
        dA = tb.t() @ h.double()
        dB = scaling * dq.t() @ u[:, p * r:(p + 1) * r].double()
        gA, gB = grad[off:off + r * D].reshape(r, D), grad[off + r * D:off + 2 * r * D].reshape(D, r)
        off += 2 * r * D
        assert maxdiff(gA, dA) < 2e-4 * max(1.0, dA.abs().max().item()), p
        assert maxdiff(gB, dB) < 2e-4 * max(1.0, dB.abs().max().item()), p
    assert float(t_out[:, 3 * r:].abs().max()) == 0


# ------------------------------------------------------------------------------------------------ GEMM epilogues
def _gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("variant", [0, 9, 10])
@pytest.mark.parametrize("M,Nn,K", [(300, 512, 128), (2740, 3072, 768)])
def test_gemm_train_epilogues(variant, M, Nn, K):
    g = torch.Generator().manual_seed(M + Nn)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(Nn, K, generator=g) * 0.05)
    b = torch.randn(Nn, generator=g)
    pre_ref = A.double() @ W.double().t() + b.double()
    lib = N.load()
    As, Ws, bs = A.to(DEV), W.to(DEV), b.to(DEV)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
    pre = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
    N.check(lib.ucod_gemm_bf16_train(N.EPI_BIAS_GELU_SAVE_BF16, N.ptr(As), N.ptr(Ws), N.ptr(out), M, Nn, K, N.ptr(bs), None, N.ptr(pre), variant,
                                     N.stream()), "gelu_save")
    assert rel_l2(pre.float(), pre_ref) < 4e-3
    assert rel_l2(out.float(), OV.gelu_erf(pre_ref)) < 5e-3
    # fc2-dgrad style: out = (A W^T) * gelu'(aux)
    aux = bf(torch.randn(M, Nn, generator=g) * 1.5)
    auxs = aux.to(DEV)
    N.check(lib.ucod_gemm_bf16_train(N.EPI_GELU_BWD_BF16, N.ptr(As), N.ptr(Ws), N.ptr(out), M, Nn, K, None, N.ptr(auxs), None, variant, N.stream()),
            "gelu_bwd")
    ref = (A.double() @ W.double().t()) * _gelu_grad(aux.double())
    assert rel_l2(out.float(), ref) < 5e-3
    # plain products (NULL bias) in both output types
    o32 = torch.empty(M, Nn, device=DEV)
    N.check(lib.ucod_gemm_bf16(N.EPI_BIAS_F32, N.ptr(As), N.ptr(Ws), N.ptr(o32), M, Nn, K, None, None, None, None, 0, variant, N.stream()), "plain f32")
    assert rel_l2(o32, A.double() @ W.double().t()) < 1e-5
    N.check(lib.ucod_gemm_bf16(N.EPI_BIAS_BF16, N.ptr(As), N.ptr(Ws), N.ptr(out), M, Nn, K, None, None, None, None, 0, variant, N.stream()), "plain bf16")
    assert rel_l2(out.float(), A.double() @ W.double().t()) < 4e-3


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, heads):
    B, T, D = q.shape
    hd = D // heads
    qh, kh, vh = (t.view(B, T, heads, hd).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(2, 3) * hd ** -0.5
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, T, D), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,tok,heads", [(1, 26, 2), (2, 200, 3), (1, 1370, 2), (2, 129, 1)])
def test_attention_backward(B, tok, heads):
    g = torch.Generator().manual_seed(B * 1000 + tok)
    D = heads * 64
    c = 0.125 * 1.4426950408889634
    q = bf(torch.randn(B, tok, D, generator=g) * 1.5)
    k = bf(torch.randn(B, tok, D, generator=g) * 1.5)
    v = bf(torch.randn(B, tok, D, generator=g))
    qs = bf(q.float() * c)                                              # what the QKV epilogue stores
    q_eff = (qs.double() / c).requires_grad_(True)                      # the unscaled q the gradient refers to
    kd, vd = k.double().requires_grad_(True), v.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(q_eff, kd, vd, heads)
    do = bf(torch.randn(B, tok, D, generator=g))
    gq, gk, gv = torch.autograd.grad((o_ref * do.double()).sum(), (q_eff, kd, vd))
    lib = N.load()
    qkv = torch.cat((qs, k, v), -1).reshape(B * tok, 3 * D).contiguous().to(DEV)
    out = torch.empty(B * tok, D, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, heads, tok, device=DEV)
    N.check(lib.ucod_attention_fwd_lse(N.ptr(qkv), N.ptr(out), N.ptr(lse), B, tok, heads, N.stream()), "fwd_lse")
    assert rel_l2(out.float().reshape(B, tok, D), o_ref) < 8e-3
    assert maxdiff(lse.cpu() / 1.4426950408889634, lse_ref.detach()) < 2e-3 * max(1.0, lse_ref.abs().max().item())
    ld = 3 * D + AUG
    dqkv = torch.full((B * tok, ld), 5.0, dtype=torch.bfloat16, device=DEV)
    delta = torch.empty(B, heads, tok, device=DEV)
    dos = do.reshape(B * tok, D).contiguous().to(DEV)
    N.check(lib.ucod_attention_bwd(N.ptr(qkv), N.ptr(out), N.ptr(dos), N.ptr(lse), N.ptr(delta), N.ptr(dqkv), ld, B, tok, heads, N.stream()), "bwd")
    d = dqkv.float().cpu().reshape(B, tok, ld)
    assert float((d[..., 3 * D:] - 5.0).abs().max()) == 0              # aug columns untouched
    for name, got, ref in (("dq", d[..., :D], gq), ("dk", d[..., D:2 * D], gk), ("dv", d[..., 2 * D:3 * D], gv)):
        assert rel_l2(got, ref) < 1.5e-2, (name, rel_l2(got, ref))


# ------------------------------------------------------------------------------------------------ whole passes
def _engine_from_golden(g):
    sd = sub(g, "sd.")
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    eng = ViTLoRAEngine(base, heads=2, r=2, lora_alpha=4, device=DEV)
    eng.load_lora_state_dict(sd)
    return eng, sd


def test_g12_end_to_end_lora_gradients():
    """HF Dinov2Model + LoRA autograd (golden) vs ucod_vit_forward_train / ucod_vit_backward.  bf16 operands through 3 layers
    forward and backward: key within 3e-2 abs (O(3) values), every LoRA gradient within 4e-2 relative L2."""
    g = load_golden("g12_lora_backbone")
    eng, sd = _engine_from_golden(g)
    key = eng.forward_train(g["x"].to(DEV))
    assert maxdiff(key.cpu(), g["key"]) < 3e-2 * max(1.0, g["key"].abs().max().item())
    eng.backward(g["dkey"].to(DEV))
    grads = eng.lora_state_dict(grads=True)
    worst = 0.0
    for k, v in grads.items():
        ref = g["grad." + k]
        if float(ref.abs().max()) == 0.0:
            assert float(v.abs().max()) == 0.0, k                       # last layer's query / value LoRA: exactly zero
            continue
        worst = max(worst, rel_l2(v, ref))
        assert rel_l2(v, ref) < 4e-2, (k, rel_l2(v, ref))
    assert worst > 0


def test_forward_train_matches_inference_forward():
    """Same weights, LoRA B = 0: the training forward's key map must equal the inference engine's bit for bit up to the
    different GEMM K extent (aug columns are zero) -- here: identical within bf16 noise, and deterministic across calls."""
    from ucod_dpl_amd.vit_engine import ViTEngine
    g = load_golden("g8_dinov2_native")
    base = sub(g, "sd.")
    inf = ViTEngine(base, heads=2, device=DEV, attn_variant=2, half="bf16")       # (the training engine is the bf16 build)
    eng = ViTLoRAEngine(base, heads=2, device=DEV)
    x = g["x"].to(DEV)
    k0 = inf(x)
    k1 = eng.forward_train(x)
    k2 = eng.forward_train(x)
    assert torch.equal(k1, k2)
    assert maxdiff(k0.cpu(), k1.cpu()) < 1e-2 * max(1.0, k0.abs().max().item())
    assert maxdiff(k1.cpu(), g["key"]) < 3e-2 * max(1.0, g["key"].abs().max().item())


@pytest.mark.parametrize("dropout", [0.0, 0.3])
def test_forward_nograd_is_the_training_forward_without_the_saved_activations(dropout):
    """ViTLoRAEngine.forward_nograd (ucod_vit_forward_lora_infer: the EMA teacher's pass, models/modules/full_model.py:84,108-111) against
    forward_train on the same engine state: non-zero LoRA B matrices, the same dropout masks (same seed and step), f32 residual stream ->
    same arithmetic up to the GEMM tile shapes; fp16 residual stream -> within the bf16 operands' own noise; against the reference's key
    map (G8, LoRA off) at the inference engine's tolerance."""
    g = load_golden("g8_dinov2_native")
    base = sub(g, "sd.")
    x = g["x"].to(DEV)

    def engine():
        e = ViTLoRAEngine(base, heads=2, device=DEV, lora_dropout=dropout, seed=11, generator=torch.Generator().manual_seed(3))
        gen = torch.Generator().manual_seed(4)
        rD = e.r * e.D
        for p in range(3):                                        # B matrices away from zero so that the LoRA branch matters
            e.lora[:, p * 2 * rD + rD:(p + 1) * 2 * rD] = (torch.randn(e.L, rD, generator=gen) * 0.05).to(DEV)
        e.repack()
        return e

    k_train = engine().forward_train(x)
    k_f32 = engine().forward_nograd(x, resid16=False)
    k_f16 = engine().forward_nograd(x, resid16=True)
    assert k_f32.shape == k_train.shape and bool(torch.isfinite(k_f16).all())
    assert rel_l2(k_f32, k_train) < 2e-3, rel_l2(k_f32, k_train)
    assert rel_l2(k_f16, k_train) < 4e-3, rel_l2(k_f16, k_train)
    e0 = ViTLoRAEngine(base, heads=2, device=DEV)                 # B = 0: the LoRA branch vanishes -> the reference's key map
    assert maxdiff(e0.forward_nograd(x).cpu(), g["key"]) < 3e-2 * max(1.0, g["key"].abs().max().item())


def test_medium_model_lora_gradients_vs_oracle_autograd():
    """D=256, 4 heads, 4 layers, 9x9 patches (82 tokens), batch 3: HIP passes vs oracle/vit.py + torch autograd (CPU, f32)."""
    from transformers import Dinov2Config, Dinov2Model
    torch.manual_seed(21)
    cfg = Dinov2Config(hidden_size=256, num_hidden_layers=4, num_attention_heads=4, image_size=126, patch_size=14, mlp_ratio=4,
                       layerscale_value=1.0)
    m = Dinov2Model(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
            if "position_embeddings" in n or "cls_token" in n:
                p.mul_(0.05)
    base = {k: v.detach() for k, v in m.state_dict().items()}
    gen = torch.Generator().manual_seed(4)
    eng = ViTLoRAEngine(base, heads=4, r=2, lora_alpha=4, device=DEV, generator=gen)
    lsd = eng.lora_state_dict()
    for k in lsd:
        if "lora_B" in k:
            lsd[k] = 0.05 * torch.randn(lsd[k].shape, generator=gen)
    eng.load_lora_state_dict(lsd)
    x = torch.randn(3, 3, 126, 126, generator=gen)
    dkey = torch.randn(3, 256, 9, 9, generator=gen)
    sd = dict(base)
    sd.update({k: v.cpu() for k, v in eng.lora_state_dict().items()})
    key_ref, gref = OV.dinov2_lora_grads(x, sd, heads=4, dkey=dkey, lora_scale=2.0)
    key = eng.forward_train(x.to(DEV))
    assert maxdiff(key.cpu(), key_ref) < 3e-2 * max(1.0, key_ref.abs().max().item())
    eng.backward(dkey.to(DEV))
    got = eng.lora_state_dict(grads=True)
    for k, ref in gref.items():
        if float(ref.abs().max()) == 0.0:
            assert float(got[k].abs().max()) == 0.0, k
        else:
            assert rel_l2(got[k], ref) < 5e-2, (k, rel_l2(got[k], ref))


def test_full_model_backward_through_decoder_and_backbone():
    """models/modules/full_model.py mirror: image -> LoRA backbone -> key hook (bilinear 68x68) -> DBA decoder -> loss, and
    loss.backward() down to the LoRA matrices, against the CPU oracle (oracle ViT + oracle decoder + torch autograd)."""
    from oracle import decoder as OD
    from oracle.resize import torch_bilinear
    from ucod_dpl_amd.engine.config import CfgNode
    from ucod_dpl_amd.models.modules.full_model import full_model, load_lora
    from ucod_dpl_amd.models.uscod import baseline
    g = load_golden("g12_lora_backbone")
    sd = sub(g, "sd.")
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    torch.manual_seed(0)
    cfg = CfgNode(dict(model_cfg=dict(dim=128, feature_size=8, ema_weight=0.99, enable_ocm=False, freeze_lora=False), lora_cfg=dict(r=2, lora_alpha=4, lora_dropout=0.0)))
    dec = baseline(cfg.model_cfg).to(DEV)
    bb = load_lora(cfg.lora_cfg, base, heads=2, device=DEV)
    bb.engine.load_lora_state_dict(sd)
    fm = full_model(cfg, bb, dec)
    fm.hook_size = HS = 8              # the tiny model has a 5x5 grid: 5->68 is outside the adjoint kernel's tap budget (37->68 is the real case)
    x = g["x"]
    gen = torch.Generator().manual_seed(9)
    r1, r2 = torch.randn(2, 1, HS, HS, generator=gen), torch.randn(2, 1, HS, HS, generator=gen)
    fg, bgm, extra = fm(x.to(DEV))
    loss = (fg * r1.to(DEV)).sum() + (bgm * r2.to(DEV)).sum() + 100.0 * extra
    loss.backward()
    got = fm.backbone.lora.grad
    assert got is not None and got.shape == bb.engine.lora.shape
    # oracle
    names = [k for k in sd if ".lora_" in k]
    leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd2 = dict(sd)
    sd2.update(leaf)
    _, key = OV.dinov2_forward(x, sd2, heads=2, full_last_layer=False, lora_scale=2.0)
    key68 = torch_bilinear(key, HS, HS)
    p = {k: v.detach().cpu() for k, v in dec.decoder.state_dict().items()}
    ofg, obg, oextra = OD.rev_decoder_forward(key68, p, orth="gram")
    oloss = (ofg * r1).sum() + (obg * r2).sum() + 100.0 * oextra
    gref = torch.autograd.grad(oloss, [leaf[k] for k in names], allow_unused=True)
    assert abs(loss.item() - oloss.item()) < 2e-2 * max(1.0, abs(oloss.item()))
    eng_grads = {}
    flat = got.detach()
    rD = 2 * 128
    for i in range(3):
        for pi, nm in enumerate(("query", "key", "value")):
            basek = f"encoder.layer.{i}.attention.attention.{nm}."
            eng_grads[basek + "lora_A.weight"] = flat[i, pi * 2 * rD:pi * 2 * rD + rD].reshape(2, 128)
            eng_grads[basek + "lora_B.weight"] = flat[i, pi * 2 * rD + rD:(pi + 1) * 2 * rD].reshape(128, 2)
    for k, ref in zip(names, gref):
        if ref is None or float(ref.abs().max()) == 0.0:
            assert float(eng_grads[k].abs().max()) == 0.0, k
        else:
            assert rel_l2(eng_grads[k], ref) < 5e-2, (k, rel_l2(eng_grads[k], ref))
    # EMA path: no grad, same call shape
    t = fm(x.to(DEV), ema=True)
    assert t.shape == (2, 1, HS, HS) and not t.requires_grad


def test_fused_full_step_against_oracle():
    """TrainLoop._process_batch_full (images -> LoRA backbone -> decoder/APM/discriminator step -> backbone backward ->
    optimisers) vs the CPU oracle: oracle ViT(+LoRA) feeding oracle process_batch, autograd down to the LoRA matrices."""
    from oracle import train_step as OT
    from test_gpu_train_step import make_cfg
    from ucod_dpl_amd.engine.runner import StandardRunner, TrainLoop
    g = load_golden("g12_lora_backbone")
    sd = sub(g, "sd.")
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    cfg = make_cfg(C=128, fs=8)
    torch.manual_seed(3)
    runner = StandardRunner(cfg)
    loop = TrainLoop(runner.config, runner)
    eng = ViTLoRAEngine(base, heads=2, r=2, lora_alpha=4, device=DEV)
    eng.load_lora_state_dict(sd)
    loop.attach_lora_backbone(eng)
    dec0 = {k: v.detach().cpu().clone() for k, v in runner.model.decoder.state_dict().items()}
    ema0 = {k: v.detach().cpu().clone() for k, v in runner.model.decoder_ema.state_dict().items()}
    disc0 = {k: v.detach().cpu().clone() for k, v in runner.discriminator.state_dict().items()}
    lora0 = eng.lora.detach().cpu().clone()
    gen = torch.Generator().manual_seed(11)
    x = g["x"]
    pl = (torch.rand(2, 1, 16, 16, generator=gen) > 0.5).float()
    loss = loop._process_batch_full(x, pl)
    # ---- oracle
    CFG = dict(feature_size=8, ema_weight=0.99, lr0=6e-4, dis_lr0=1e-3, step_lr_size=2, step_lr_gamma=0.95, dis_step_lr_size=2,
               dis_step_lr_gamma=0.95, max_epoch=25, start_finetune=-5)
    st = OT.TrainState(dec0, ema0, disc0, CFG)
    names = [k for k in sd if ".lora_" in k]
    leaf = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd2 = dict(sd)
    sd2.update(leaf)
    _, key = OV.dinov2_forward(x, sd2, heads=2, full_last_layer=False, lora_scale=2.0)
    out = OT.process_batch(st, key, pl, orth="gram", extra_leaves=[leaf[k] for k in names])
    assert abs(loss.item() - out["loss"].item()) < 2e-2 * max(1.0, abs(out["loss"].item())), (loss.item(), out["loss"].item())
    got = eng.lora_state_dict(grads=True)
    checked = 0
    for k, ref in zip(names, out["extra_grads"]):
        if float(ref.abs().max()) == 0.0:
            assert float(got[k].abs().max()) == 0.0, k
        else:
            assert rel_l2(got[k], ref) < 8e-2, (k, rel_l2(got[k], ref))
            checked += 1
    assert checked == 14                                                # 3 layers x 3 x 2 minus the last layer's query/value
    # the LoRA parameters moved (AdamW), the teacher's copy followed by EMA (alpha = 0 on the first step: copy of the student)
    assert float((eng.lora.cpu() - lora0).abs().max()) > 0
    assert maxdiff(loop.lora_engine_ema.lora.cpu(), eng.lora.cpu()) < 1e-7
    # decoder parameters after the step vs the oracle's optimiser
    sd_after = {k: v.cpu() for k, v in runner.model.decoder.state_dict().items()}
    for k, v in st.dec.items():
        if k == "learnable_embedding":
            continue
        assert maxdiff(sd_after[k], v) < 2e-3 * max(1e-3, v.abs().max().item()) + 1.5e-3, (k, maxdiff(sd_after[k], v))


def test_lora_dropout_masks_and_gradients():
    """LoRA dropout (LoraConfig.lora_dropout): counter-based masks regenerated by every kernel.  (1) the forward's masked
    down-projection against numpy masks from the documented hash; (2) whole forward/backward with p = 0.3 against the oracle given
    the SAME masks; (3) keep rate; (4) eval() switches it off and reproduces the no-dropout key map."""
    g = load_golden("g12_lora_backbone")
    sd = sub(g, "sd.")
    base = {k: v for k, v in sd.items() if ".lora_" not in k}
    p_drop, seed = 0.3, 1234
    eng = ViTLoRAEngine(base, heads=2, r=2, lora_alpha=4, device=DEV, lora_dropout=p_drop, seed=seed)
    eng.train_streams = 1
    eng.load_lora_state_dict(sd)
    x, dkey = g["x"], g["dkey"]
    key = eng.forward_train(x.to(DEV))
    step_seed = eng._step_seed
    rows, D = 2 * 26, 128
    masks = {(i, nm): OV.lora_dropout_mask(step_seed, i, pi, rows, D, p_drop) for i in range(3) for pi, nm in enumerate(("query", "key", "value"))}
    m0 = masks[(0, "query")]
    p_eff = math.floor(p_drop * 1024) / 1024                    # ABI 3: a 10-bit field of ONE mix per element decides each projection
    assert abs(float((m0 > 0).float().mean()) - (1 - p_eff)) < 0.03 and abs(float(m0.max()) - 1 / (1 - p_eff)) < 1e-6
    assert abs(float(m0.mean()) - 1.0) < 0.05                    # unbiased: E[mask] = 1
    assert not torch.equal(masks[(0, "query")], masks[(0, "key")]) and not torch.equal(masks[(0, "query")], masks[(1, "query")])
    # the three projections share the mix but not the bits: their keep decisions are independent (correlation of the indicators ~ 0)
    kq, kk, kv = ((masks[(0, n)] > 0).float().flatten() for n in ("query", "key", "value"))
    for a_, b_ in ((kq, kk), (kq, kv), (kk, kv)):
        assert abs(float(torch.corrcoef(torch.stack((a_, b_)))[0, 1])) < 0.05
    key_ref, gref = OV.dinov2_lora_grads(x, sd, heads=2, dkey=dkey, lora_scale=2.0, lora_masks=masks)
    assert maxdiff(key.cpu(), key_ref) < 3e-2 * max(1.0, key_ref.abs().max().item())
    eng.backward(dkey.to(DEV))
    got = eng.lora_state_dict(grads=True)
    for k, ref in gref.items():
        if float(ref.abs().max()) == 0.0:
            assert float(got[k].abs().max()) == 0.0, k
        else:
            assert rel_l2(got[k], ref) < 5e-2, (k, rel_l2(got[k], ref))
    # the dropped branch really changes the answer: without masks the oracle disagrees
    _, g_nomask = OV.dinov2_lora_grads(x, sd, heads=2, dkey=dkey, lora_scale=2.0)
    k0 = "encoder.layer.0.attention.attention.query.lora_A.weight"
    assert rel_l2(got[k0], g_nomask[k0]) > 0.1
    # a new step draws new masks; eval() turns dropout off
    key2 = eng.forward_train(x.to(DEV))
    assert eng._step_seed != step_seed and not torch.equal(key2, key)
    eng.eval()
    k_eval = eng.forward_train(x.to(DEV))
    ref_eng = ViTLoRAEngine(base, heads=2, r=2, lora_alpha=4, device=DEV)
    ref_eng.train_streams = 1
    ref_eng.load_lora_state_dict(sd)
    assert torch.equal(k_eval, ref_eng.forward_train(x.to(DEV)))
    eng.backward(dkey.to(DEV))
    ref_eng.backward(dkey.to(DEV))
    assert torch.equal(eng.lora_grad, ref_eng.lora_grad)
