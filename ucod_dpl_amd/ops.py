"""Functional wrappers over the C ABI: allocate outputs with torch, launch on the current stream.

Every function requires CUDA (ROCm) tensors and raises otherwise -- there is no CPU path here.
"""
import ctypes as C
import os

import torch

from . import native as N
from .native import ptr, stream, check


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    return t


def _bf16(t):
    if t.dtype != torch.bfloat16:
        raise TypeError(f"expected bfloat16, got {t.dtype}")
    return t


# ----------------------------------------------------------------------------- backbone pieces
def gemm_bf16(epilogue, A, Bm, out, M, Nn, K, bias=None, scale=None, resid=None, pos=None, tok=0, variant=0):
    """``variant`` outside the product set (native.GEMM_PRODUCT_VARIANTS) goes to the laboratory library (`make variants`)."""
    if variant in N.GEMM_PRODUCT_VARIANTS:
        fn, what = N.load().ucod_gemm_bf16, "ucod_gemm_bf16"
    else:
        fn, what = N.load_lab().ucod_gemm_bf16_lab, "ucod_gemm_bf16_lab"
    check(fn(epilogue, ptr(_bf16(A)), ptr(_bf16(Bm)), ptr(out), M, Nn, K, ptr(bias), ptr(scale), ptr(resid), ptr(pos), tok, variant, stream()), what)
    return out


def linear_bf16(x, w, b, gelu=False, variant=0):
    """x bf16 [M,K], w bf16 [N,K], b f32 [N] -> bf16 [M,N]."""
    M, K = x.shape
    out = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=x.device)
    return gemm_bf16(N.EPI_BIAS_GELU_BF16 if gelu else N.EPI_BIAS_BF16, x, w, out, M, w.shape[0], K, bias=_f32(b), variant=variant)


def linear_bf16_asm(x, w, b, out=None, dbg=None, form=0):
    """LABORATORY: the hand-placed persistent GEMM (variants/gemm_asm_lab.hip; K = 768, N % 256 == 0): x bf16 [M,K], w bf16 [N,K], b f32 [N] -> bf16 [M,N]."""
    M, K = x.shape
    out = torch.empty(M, w.shape[0], dtype=torch.bfloat16, device=x.device) if out is None else out
    check(N.load_lab().ucod_gemm_bf16_asm_lab(ptr(_bf16(x)), ptr(_bf16(w)), ptr(_f32(b)), ptr(out), M, w.shape[0], K, form, ptr(dbg), stream()), "ucod_gemm_bf16_asm_lab")
    return out


def linear_scale_resid(x, w, b, scale, resid, variant=0):
    """resid f32 [M,N] + scale[n]*(x w^T + b) -> new f32 [M,N]."""
    M, K = x.shape
    out = torch.empty_like(resid)
    return gemm_bf16(N.EPI_BIAS_SCALE_RESID_F32, x, w, out, M, w.shape[0], K, bias=b, scale=scale, resid=resid, variant=variant)


# ---- split-operand (f32-equivalent) pieces: include/ucod_dpl.h, "split-operand backbone pass"; csrc/split.hip
def split_products(terms):
    n = N.load().ucod_split_products(int(terms))
    if n == 0:
        raise ValueError(f"terms must be 2 or 3, got {terms}")
    return n


def split_rows(x, terms, role, op=0, alpha=1.0):
    """x f32 [M,K] (rows may be strided) -> bf16 [M, P K]: segment p holds term {0,0,1,1,0,2}[p] (role 0, the A side) / {0,1,0,1,2,0}[p] (role 1, the B side)
    of x = x0 + x1 (+ x2).  op 1: exact-erf GELU of x first; op 2: x * alpha first."""
    _f32(x)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("split_rows takes a 2-d tensor with contiguous rows")
    if not x.is_cuda:
        raise RuntimeError("ucod_dpl_amd: tensor is not on a GPU; the HIP path has no CPU fallback")
    M, K = x.shape
    out = torch.empty(M, split_products(terms) * K, dtype=torch.bfloat16, device=x.device)
    check(N.load().ucod_split_rows(x.data_ptr(), x.stride(0), ptr(out), M, K, int(terms), int(role), int(op), float(alpha), stream()), "ucod_split_rows")
    return out


def linear_split(x, w, b, terms, variant=0):
    """f32-equivalent x w^T + b on the bf16 matrix pipe: x f32 [M,K], w f32 [N,K], b f32 [N] -> f32 [M,N] (ucod_gemm_bf16 over the K-concatenated split operands)."""
    M, K = x.shape
    P = split_products(terms)
    out = torch.empty(M, w.shape[0], dtype=torch.float32, device=x.device)
    return gemm_bf16(N.EPI_BIAS_F32, split_rows(x, terms, 0), split_rows(w, terms, 1), out, M, w.shape[0], P * K, bias=_f32(b), variant=variant)


def layernorm_split(x, gamma, beta, eps, terms, role=0):
    rows, D = x.shape
    out = torch.empty(rows, split_products(terms) * D, dtype=torch.bfloat16, device=x.device)
    check(N.load().ucod_layernorm_split(ptr(_f32(x)), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(out), rows, D, float(eps), int(terms), int(role), stream()), "ucod_layernorm_split")
    return out


def attention_split(qkv, B, tok, heads, terms, qscale=0.125 * 1.4426950408889634):
    """qkv f32 [B*tok, 3*heads*64] -> the attention output as the A-side split operand bf16 [B*tok, P*heads*64] (softmax(Q K^T / 8) V, f32-equivalent)."""
    lib = N.load()
    need = lib.ucod_attention_split_operand_bytes(B, tok, heads, int(terms))
    if need == 0:
        raise ValueError("unsupported attention geometry")
    opnd = torch.empty(need, dtype=torch.uint8, device=qkv.device)
    out = torch.empty(B * tok, split_products(terms) * heads * 64, dtype=torch.bfloat16, device=qkv.device)
    check(lib.ucod_qkv_split(ptr(_f32(qkv)), ptr(opnd), B, tok, heads, int(terms), float(qscale), stream()), "ucod_qkv_split")
    check(lib.ucod_attention_split_fwd(ptr(opnd), ptr(out), B, tok, heads, int(terms), stream()), "ucod_attention_split_fwd")
    return out


def unsplit(xs, terms, role, K):
    """The f32 value a split operand [M, P K] stands for (sum of its distinct terms): test helper, torch arithmetic on the device."""
    P = split_products(terms)
    seg = xs.view(xs.shape[0], P, K).float()
    order = ([0, 0, 1, 1, 0, 2] if role == 0 else [0, 1, 0, 1, 2, 0])[:P]
    first = [order.index(t) for t in range(int(terms))]
    return sum(seg[:, i] for i in first)


def row_stats_h16(x, eps):
    """x fp16 [rows, D] (the fp16 residual stream) -> f32 [rows, 2] = (rstd, -mean * rstd): what the LayerNorm-folded epilogues read."""
    if x.dtype != torch.float16:
        raise TypeError(f"expected float16, got {x.dtype}")
    rows, D = x.shape
    st = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    check(N.load("f16").ucod_row_stats_h16(ptr(x), ptr(st), rows, D, float(eps), stream()), "ucod_row_stats_h16")
    return st


def fold_layernorm_linear(gamma, beta, w, b):
    """(w_folded fp16 [N,K], bias_folded f32 [N], colsum f32 [N]) of LayerNorm(gamma, beta) -> Linear(w, b): include/ucod_dpl.h, ucod_gemm_lnfold."""
    from .fold import fold_layernorm_linear as _fold
    return _fold(_f32(gamma), _f32(beta), _f32(w), _f32(b))


def linear_lnfold(x, stats, wf, bias_f, colsum, gelu=False, scale=None, variant=0, partials=None, eps=1e-6):
    """LayerNorm + Linear (+ GELU | * scale) as ONE GEMM on the un-normalised fp16 rows (fp16-operand build): x fp16 [M,K] -> fp16 [M,N].
    ``stats`` f32 [M,2] (row_stats_h16) or ``partials`` f32 [M,nslot,2] (left by linear_scale_resid_h16_stats / the patch embedding)."""
    if x.dtype != torch.float16 or wf.dtype != torch.float16:
        raise TypeError("linear_lnfold takes fp16 rows and fp16 folded weights")
    M, K = x.shape
    out = torch.empty(M, wf.shape[0], dtype=torch.float16, device=x.device)
    nslot = 0 if partials is None else partials.shape[1]
    check(N.load("f16").ucod_gemm_lnfold(N.EPI_LNFOLD_GELU_BF16 if gelu else N.EPI_LNFOLD_BIAS_BF16, ptr(x), ptr(wf), ptr(out), M, wf.shape[0], K,
                                         ptr(_f32(bias_f)), ptr(_f32(colsum)), ptr(stats), ptr(partials), nslot, float(eps), ptr(scale), variant, stream()),
          "ucod_gemm_lnfold")
    return out


def linear_scale_resid_h16_stats(a, w, b, scale, resid, half="f16"):
    """resid f16 [M,N] + scale[n] * (a w^T + b) -> (new f16 [M,N], partials f32 [M, N/64, 2]) -- the out-projection / fc2 epilogue on the fp16 stream
    that also leaves each row's (sum, sum of squares) per 64-column slot.  Raises for shapes the large-tile kernels do not take."""
    M, K = a.shape
    Nn = w.shape[0]
    out = torch.empty(M, Nn, dtype=torch.float16, device=a.device)
    part = torch.empty(M, Nn // 64, 2, dtype=torch.float32, device=a.device)
    check(N.load(half).ucod_gemm_bf16_stats(N.EPI_BIAS_SCALE_RESID_H16_STATS, ptr(a), ptr(w), ptr(out), M, Nn, K, ptr(_f32(b)), ptr(_f32(scale)), ptr(resid), None, 0,
                                            ptr(part), Nn // 64, stream()), "ucod_gemm_bf16_stats")
    return out, part


def layernorm(x, gamma, beta, eps, out_f32=False):
    rows, D = x.shape
    y = torch.empty(rows, D, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    check(N.load().ucod_layernorm(ptr(_f32(x)), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(y), rows, D, float(eps), int(out_f32), stream()),
          "ucod_layernorm")
    return y


def attention(qkv, B, tok, heads, scale=0.125, variant=0):
    """``scale`` != 0: generic kernel; ``scale`` == 0: Q pre-scaled by hd^-1/2 log2 e (the product kernel).  ``variant`` outside
    native.ATTN_PRODUCT_VARIANTS selects a laboratory kernel of libucod_dpl_variants.so (`make variants`)."""
    out = torch.empty(B * tok, heads * 64, dtype=torch.bfloat16, device=qkv.device)
    if variant in N.ATTN_PRODUCT_VARIANTS:
        check(N.load().ucod_attention_fwd(ptr(_bf16(qkv)), ptr(out), B, tok, heads, float(scale), variant, stream()), "ucod_attention_fwd")
    else:
        check(N.load_lab().ucod_attention_fwd_lab(ptr(_bf16(qkv)), ptr(out), B, tok, heads, float(scale), variant, stream()), "ucod_attention_fwd_lab")
    return out


def attention_asm(qkv, B, tok, heads, form=0, want_lse=False):
    """LABORATORY (libucod_dpl_variants.so): the hand-placed gfx950 assembly kernels of round 4 -- form 0 = ucod_attn_fwd_pw64 (4 waves x 64 query rows),
    1 = ucod_attn_fwd_pw32 (8 waves x 32 rows).  Q pre-scaled; raises for fewer than three key tiles.  -> out [, lse [B, heads, tok]]"""
    out = torch.empty(B * tok, heads * 64, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, heads, tok, dtype=torch.float32, device=qkv.device) if want_lse else None
    check(N.load_lab().ucod_attention_fwd_asm_lab(ptr(_bf16(qkv)), ptr(out), ptr(lse), B, tok, heads, form, stream()), "ucod_attention_fwd_asm_lab")
    return (out, lse) if want_lse else out


def attention_fp8(qkv, B, tok, heads, q_exp=5, k_exp=3, v_exp=3):
    """The fp8 (e4m3, block-scaled MFMA) attention path of BASELINE configs[4]; ``qkv`` bf16 with Q pre-scaled by hd^-1/2 * log2 e."""
    lib = N.load()
    out = torch.empty(B * tok, heads * 64, dtype=torch.bfloat16, device=qkv.device)
    ws = torch.empty(lib.ucod_attention_fp8_workspace_bytes(B, tok, heads), dtype=torch.uint8, device=qkv.device)
    check(lib.ucod_attention_fwd_fp8(ptr(_bf16(qkv)), ptr(out), ptr(ws), ws.numel(), B, tok, heads, q_exp, k_exp, v_exp, stream()),
          "ucod_attention_fwd_fp8")
    return out


def qkv_fp8_attention(h, w_qkv, b_qkv, B, tok, heads, q_exp=5, k_exp=3, v_exp=3):
    """Fused fp8 path: QKV projection with the e4m3 epilogue (UCOD_EPI_QKV_FP8) straight into Q8 | K8 | V8, then the fp8 attention
    kernel that transposes V on the fly.  ``h`` bf16 [B*tok, D] (LayerNorm output), ``w_qkv`` bf16 [3D, D], ``b_qkv`` f32 [3D]."""
    import math
    lib = N.load()
    D = heads * 64
    M = B * tok
    ws = torch.empty(lib.ucod_attention_fp8_workspace_bytes(B, tok, heads), dtype=torch.uint8, device=h.device)
    scale = torch.empty(3 * D, dtype=torch.float32, device=h.device)
    check(lib.ucod_fill_qscale3(ptr(scale), D, 0.125 * math.log2(math.e) * 2.0 ** q_exp, 2.0 ** k_exp, 2.0 ** v_exp, stream()), "ucod_fill_qscale3")
    check(lib.ucod_attention_fp8_zero_pad(ptr(ws), B, tok, heads, stream()), "ucod_attention_fp8_zero_pad")
    check(lib.ucod_gemm_bf16(N.EPI_QKV_FP8, ptr(_bf16(h)), ptr(_bf16(w_qkv)), ptr(ws), M, 3 * D, D, ptr(_f32(b_qkv)), ptr(scale), None, None, tok, 0,
                             stream()), "ucod_gemm_bf16(QKV_FP8)")
    out = torch.empty(M, D, dtype=torch.bfloat16, device=h.device)
    check(lib.ucod_attention_fwd_fp8_fused(ptr(ws), ptr(out), B, tok, heads, q_exp, k_exp, v_exp, stream()), "ucod_attention_fwd_fp8_fused")
    return out, ws


def patch_im2col(img, P, Kpad):
    B, Cc, H, W = img.shape
    out = torch.empty(B * (H // P) * (W // P), Kpad, dtype=torch.bfloat16, device=img.device)
    check(N.load().ucod_patch_im2col(ptr(_f32(img)), ptr(out), B, Cc, H, W, P, Kpad, stream()), "ucod_patch_im2col")
    return out


def cast_bf16(t, lib=None):
    """f32 -> the 16-bit operand type of ``lib`` (bf16 for the default library, fp16 for the f16 build; round-to-nearest-even)."""
    lib = lib or N.load()
    t = _f32(t).contiguous()
    out = torch.empty(t.shape, dtype=torch.float16 if lib.ucod_half_name() == b"f16" else torch.bfloat16, device=t.device)
    check(lib.ucod_cast_f32_bf16(ptr(t), ptr(out), t.numel(), stream()), "ucod_cast_f32_bf16")
    return out


# ----------------------------------------------------------------------------- decoder pieces
def bilinear_resize(x, oh, ow):
    x = _f32(x).contiguous()
    *lead, ih, iw = x.shape
    planes = 1
    for v in lead:
        planes *= v
    out = torch.empty(*lead, oh, ow, dtype=torch.float32, device=x.device)
    check(N.load().ucod_bilinear_resize(ptr(x), ptr(out), planes, ih, iw, oh, ow, stream()), "ucod_bilinear_resize")
    return out


def bilinear_resize_adjoint(gout, ih, iw):
    gout = _f32(gout).contiguous()
    *lead, oh, ow = gout.shape
    planes = 1
    for v in lead:
        planes *= v
    gin = torch.empty(*lead, ih, iw, dtype=torch.float32, device=gout.device)
    check(N.load().ucod_bilinear_resize_adjoint(ptr(gout), ptr(gin), planes, ih, iw, oh, ow, stream()), "ucod_bilinear_resize_adjoint")
    return gin


class _BilinearResizeFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=False) with its exact transpose as the backward (both HIP kernels)."""

    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.in_hw = tuple(x.shape[-2:])
        return bilinear_resize(x, oh, ow)

    @staticmethod
    def backward(ctx, g):
        return bilinear_resize_adjoint(g.contiguous(), *ctx.in_hw), None, None


def bilinear_resize_autograd(x, oh, ow):
    return _BilinearResizeFn.apply(x, oh, ow)


_EXACT_F32 = bool(os.environ.get("UCOD_DBA_EXACT_F32"))          # the v_mfma_f32_32x32x2_f32 forms of the two decoder GEMMs


def dba_project(x, W, bias, exact=None):
    """x [B,C,H,W] f32, W [Nout,C], bias [Nout] -> d [B,Nout,HW].  Default: the f32-equivalent three-way bf16 split on the bf16 matrix
    pipe (csrc/gemm_split.hip); exact=True (or UCOD_DBA_EXACT_F32=1, or a shape the split kernel does not take): the f32 MFMA kernel."""
    B, Cc, H, Wd = x.shape
    Nout = W.shape[0]
    d = torch.empty(B, Nout, H * Wd, dtype=torch.float32, device=x.device)
    lib = N.load()
    if exact is None and _EXACT_F32:
        exact = True
    if exact is False and not (Cc % 16 == 0 and Nout in (128, 256)):
        raise ValueError("dba_project(exact=False): the split kernel needs C % 16 == 0 and Nout in (128, 256)")
    # (the split kernel runs one 192-pixel tile per workgroup, one workgroup per CU: below ~100 workgroups the f32 kernel's finer grid is faster)
    if exact is False or (not exact and Cc % 16 == 0 and Nout in (128, 256) and B * ((H * Wd + 191) // 192) >= 100):
        nb = lib.ucod_dba_project_split_workspace_bytes(Cc, Nout)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        check(lib.ucod_dba_project_split(ptr(_f32(x)), ptr(_f32(W)), ptr(_f32(bias)), ptr(d), ptr(ws), nb, B, Cc, H * Wd, Nout, stream()),
              "ucod_dba_project_split")
        return d
    check(lib.ucod_dba_project(ptr(_f32(x)), ptr(_f32(W)), ptr(_f32(bias)), ptr(d), B, Cc, H * Wd, Nout, stream()), "ucod_dba_project")
    return d


def dba_colnorm(d, c0, emb):
    B, ld_c, HW = d.shape
    norm = torch.empty(B, 128, dtype=torch.float32, device=d.device)
    check(N.load().ucod_dba_colnorm(ptr(d), ld_c, c0, ptr(_f32(emb)), ptr(norm), B, HW, stream()), "ucod_dba_colnorm")
    return norm


def dba_heads(d, c0, emb, norm, head_w, head_b, want_bg=True, want_sdiag=False, sdiag=None):
    B, ld_c, HW = d.shape
    fg = torch.empty(B, HW, dtype=torch.float32, device=d.device)
    bg = torch.empty(B, HW, dtype=torch.float32, device=d.device) if want_bg else None
    if sdiag is None:
        sdiag = torch.empty(B, dtype=torch.float32, device=d.device) if want_sdiag else None
    check(N.load().ucod_dba_heads_fwd(ptr(d), ld_c, c0, ptr(emb), ptr(norm), ptr(_f32(head_w)), ptr(_f32(head_b)), ptr(fg), ptr(bg),
                                      ptr(sdiag), B, HW, stream()), "ucod_dba_heads_fwd")
    return fg, bg, sdiag


def orth_gram(d, c0, emb, norm, sdiag):
    B, ld_c, HW = d.shape
    lib = N.load()
    ws = torch.empty(lib.ucod_orth_workspace_bytes(B, HW), dtype=torch.uint8, device=d.device)
    gram = torch.empty(B, 2, 64, 64, dtype=torch.float32, device=d.device)
    loss = torch.empty(1, dtype=torch.float32, device=d.device)
    check(lib.ucod_orth_gram_fwd(ptr(d), ld_c, c0, ptr(emb), ptr(norm), ptr(sdiag), ptr(gram), ptr(loss), ptr(ws), B, HW, stream()),
          "ucod_orth_gram_fwd")
    return loss, gram


def dba_bwd(d, c0, emb, norm, head_w, gram, gfg, gbg, gextra, g_head_w=None, g_head_b=None, g_dec_bias=None):
    B, ld_c, HW = d.shape
    lib = N.load()
    dev = d.device
    ws = torch.empty(lib.ucod_dba_bwd_workspace_bytes(B, HW), dtype=torch.uint8, device=dev)
    gd = torch.empty(B, 128, HW, dtype=torch.float32, device=dev)
    g_head_w = torch.empty(2, 64, dtype=torch.float32, device=dev) if g_head_w is None else g_head_w
    g_head_b = torch.empty(2, dtype=torch.float32, device=dev) if g_head_b is None else g_head_b
    g_dec_bias = torch.empty(128, dtype=torch.float32, device=dev) if g_dec_bias is None else g_dec_bias
    check(lib.ucod_dba_bwd(ptr(d), ld_c, c0, ptr(emb), ptr(norm), ptr(head_w), ptr(gram), ptr(_f32(gfg)), ptr(_f32(gbg)), float(gextra),
                           ptr(gd), ptr(g_head_w), ptr(g_head_b), ptr(g_dec_bias), ptr(ws), B, HW, stream()), "ucod_dba_bwd")
    return gd, g_head_w, g_head_b, g_dec_bias


def dba_wgrad(gd, x, gW=None, exact=None):
    """gW [128,C] = sum_{b,p} gd[b,n,p] x[b,c,p].  Default on large batches: the three-way bf16 split on the bf16 matrix pipe
    (csrc/gemm_split.hip, f32-equivalent); exact=True / UCOD_DBA_EXACT_F32=1 / small batches: the f32 MFMA kernel."""
    B, Cc = x.shape[0], x.shape[1]
    HW = gd.shape[2]
    gW = torch.empty(128, Cc, dtype=torch.float32, device=x.device) if gW is None else gW
    if exact is None and _EXACT_F32:
        exact = True
    if exact is False or (not exact and B * ((Cc + 383) // 384) >= 32):
        check(N.load().ucod_dba_wgrad_split(ptr(gd), ptr(_f32(x)), ptr(gW), B, Cc, HW, stream()), "ucod_dba_wgrad_split")
    else:
        check(N.load().ucod_dba_wgrad(ptr(gd), ptr(_f32(x)), ptr(gW), B, Cc, HW, stream()), "ucod_dba_wgrad")
    return gW


# ----------------------------------------------------------------------------- discriminator / APM / optimiser
def disc_params_struct(tensors):
    """tensors: dict with the 17 DiscParams field names -> CUDA f32 tensors."""
    s = N.DiscParams()
    for name, _ in N.DiscParams._fields_:
        if name == "nbt":                                   # int64[3] view of the three num_batches_tracked counters, or absent
            t = tensors.get("nbt")
            s.nbt = None if t is None else t.data_ptr()
            continue
        setattr(s, name, ptr(_f32(tensors[name])))
    return s


def step_loss(losses, extra, finetune):
    out = torch.empty(1, dtype=torch.float32, device=losses.device)
    check(N.load().ucod_step_loss(ptr(losses), ptr(extra), int(bool(finetune)), ptr(out), stream()), "ucod_step_loss")
    return out[0]


def disc_fwd(mask, tensors, update_running=True, saved=None):
    B, _, fs, _ = mask.shape
    lib = N.load()
    if saved is None:
        saved = torch.empty(lib.ucod_disc_saved_bytes(B, fs), dtype=torch.uint8, device=mask.device)
    prob = torch.empty(B, dtype=torch.float32, device=mask.device)
    ps = disc_params_struct(tensors)
    check(lib.ucod_disc_fwd(ptr(_f32(mask)), C.byref(ps), ptr(prob), ptr(saved), B, fs, int(update_running), stream()), "ucod_disc_fwd")
    return prob, saved


def disc_bce(probs_student, probs_pseudo, inv):
    """nn.BCELoss(cat(student, pseudo), [0..0, 1..1]) (mean over 2B) and inv * d(sum)/dprob, one launch -> (g_student, g_pseudo, loss 0-d)"""
    B = probs_student.numel()
    gs, gp = torch.empty_like(probs_student), torch.empty_like(probs_pseudo)
    loss = torch.empty(1, dtype=torch.float32, device=probs_student.device)
    check(N.load().ucod_disc_bce(ptr(_f32(probs_student)), ptr(_f32(probs_pseudo)), ptr(gs), ptr(gp), ptr(loss), B, float(inv), stream()), "ucod_disc_bce")
    return gs, gp, loss[0]


def copy_segments(pairs):
    """[(dst, src), ...] (at most four contiguous f32 tensors each way, same numel pairwise): one launch instead of one copy kernel per pair"""
    k = len(pairs)
    for d, s_ in pairs:
        if d.dtype != torch.float32 or s_.dtype != torch.float32 or not d.is_contiguous() or not s_.is_contiguous() or d.numel() != s_.numel():
            raise ValueError("copy_segments: contiguous f32 tensors of equal size")
    dst = (C.c_void_p * k)(*[d.data_ptr() for d, _ in pairs])
    src = (C.c_void_p * k)(*[s_.data_ptr() for _, s_ in pairs])
    n = (C.c_size_t * k)(*[d.numel() for d, _ in pairs])
    check(N.load().ucod_copy_segments(dst, src, n, k, stream()), "ucod_copy_segments")


def zero_segments(tensors):
    """Zero up to eight contiguous device tensors with ONE launch (ucod_zero_segments)."""
    k = len(tensors)
    for t in tensors:
        if not t.is_cuda or not t.is_contiguous() or (t.numel() * t.element_size()) % 4:
            raise ValueError("zero_segments: contiguous device tensors of a multiple of 4 bytes")
    dst = (C.c_void_p * k)(*[t.data_ptr() for t in tensors])
    n = (C.c_size_t * k)(*[t.numel() * t.element_size() for t in tensors])
    check(N.load().ucod_zero_segments(dst, n, k, stream()), "ucod_zero_segments")


class prezeroed:
    """Context: the accumulating outputs of the calls inside were zeroed by the caller (``zero_segments``); the entry points skip their own memsets
    (include/ucod_dpl.h: ucod_accumulators_prezeroed)."""

    def __enter__(self):
        N.load().ucod_accumulators_prezeroed(1)

    def __exit__(self, *exc):
        N.load().ucod_accumulators_prezeroed(0)
        return False


def binarize(x, logits):
    x = _f32(x).contiguous()
    out = torch.empty_like(x)
    check(N.load().ucod_binarize(ptr(x), ptr(out), x.numel(), int(logits), stream()), "ucod_binarize")
    return out


def apm_bce(pl, teacher, fg, bg, p_s, p_p, epoch_frac, gscale=1.0, losses=None):
    B = pl.shape[0]
    HW = pl.numel() // B
    dev = pl.device
    w = torch.empty(B, dtype=torch.float32, device=dev)
    merged = torch.empty(B, HW, dtype=torch.float32, device=dev)
    gfg = torch.empty(B, HW, dtype=torch.float32, device=dev)
    gbg = torch.empty(B, HW, dtype=torch.float32, device=dev)
    losses = torch.empty(4, dtype=torch.float32, device=dev) if losses is None else losses
    check(N.load().ucod_apm_bce(ptr(pl), ptr(teacher), ptr(fg), ptr(bg), ptr(p_s), ptr(p_p), float(epoch_frac), float(gscale), ptr(w),
                                ptr(merged), ptr(gfg), ptr(gbg), ptr(losses), B, HW, stream()), "ucod_apm_bce")
    return w, merged, gfg, gbg, losses


def adamw_ema(p, g, m, v, ema, lr, step, ema_alpha=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    check(N.load().ucod_adamw_ema(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), float(lr), beta1, beta2, eps, weight_decay, int(step),
                                  float(ema_alpha), stream()), "ucod_adamw_ema")


DISC_GRAD_SHAPES = (("w1", (32, 1, 3, 3)), ("g1", (32,)), ("b1", (32,)), ("w2", (16, 32, 3, 3)), ("g2", (16,)), ("b2", (16,)),
                    ("w3", (8, 16, 3, 3)), ("g3", (8,)), ("b3", (8,)), ("lin_w", None), ("lin_b", (1,)))


def disc_bwd(mask, tensors, saved, gprob, grads=None, accumulate=False):
    """Gradients of sum_b gprob[b]*prob[b] w.r.t. the 11 discriminator parameters (reference order)."""
    B, _, fs, _ = mask.shape
    lib = N.load()
    dev = mask.device
    if grads is None:
        grads = [torch.zeros(tensors["lin_w"].shape if shp is None else shp, dtype=torch.float32, device=dev) for _, shp in DISC_GRAD_SHAPES]
        accumulate = True                                   # freshly zeroed
    gs = N.DiscGrads()
    for (name, _), t in zip(DISC_GRAD_SHAPES, grads):
        setattr(gs, name, ptr(_f32(t)))
    ws = torch.empty(lib.ucod_disc_bwd_workspace_bytes(B, fs), dtype=torch.uint8, device=dev)
    ps = disc_params_struct(tensors)
    check(lib.ucod_disc_bwd(ptr(_f32(mask)), C.byref(ps), ptr(saved), ptr(_f32(gprob)), C.byref(gs), int(accumulate), ptr(ws), B, fs,
                            stream()), "ucod_disc_bwd")
    return grads


# ----------------------------------------------------------------------------- COD measures (row N4)
def cod_metrics(pred, gt):
    """pred, gt f32 [B,H,W] -> f64 [B, COD_RECORD] per-image measures (layout in include/ucod_dpl.h)."""
    pred, gt = _f32(pred).contiguous(), _f32(gt).contiguous()
    B, H, W = pred.shape
    if gt.shape != pred.shape:
        raise ValueError(f"prediction {tuple(pred.shape)} and ground truth {tuple(gt.shape)} differ")
    lib = N.load()
    nbytes = lib.ucod_cod_metrics_workspace_bytes(B, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
    out = torch.empty(B, N.COD_RECORD, dtype=torch.float64, device=pred.device)
    check(lib.ucod_cod_metrics(ptr(pred), ptr(gt), B, H, W, ptr(out), ptr(ws), nbytes, stream()), "ucod_cod_metrics")
    return out
