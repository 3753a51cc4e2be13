"""Frozen ViT backbone engine: owns bf16/f32 device copies of the weights, the pointer table the C driver
walks, and a reusable workspace.  Accepts both checkpoint layouts the reference can meet:

* HuggingFace ``Dinov2Model`` / ``ViTModel`` state dicts (data/utils/feature_extractor.py:20,25) with
  ``attention.attention.{query,key,value}`` or (transformers>=5 ViT) ``attention.{q,k,v}_proj`` names;
* the in-repo DINO ``VisionTransformer`` (models/backbones/dino.py) with fused ``attn.qkv``.

Only what the key hook needs is computed by default (feature_extractor.py:46-47,55-58); see
``ucod_vit_forward`` in include/ucod_dpl.h.
"""
import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from . import native as N
from . import ops


def _interp_pos_dinov2(pos, gh, gw):
    """transformers modeling_dinov2.py:57-95 (bicubic, size=, align_corners=False, computed in f32)."""
    n0 = pos.shape[1] - 1
    if n0 == gh * gw and gh == gw:
        return pos
    s = int(n0 ** 0.5)
    pp = pos[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2).float()
    pp = F.interpolate(pp, size=(gh, gw), mode="bicubic", align_corners=False)
    return torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)), 1)


def _interp_pos_dinov1(pos, gh, gw):
    """models/backbones/dino.py:202-221 (scale_factor + 0.1 trick)."""
    n0 = pos.shape[1] - 1
    if n0 == gh * gw and gh == gw:
        return pos
    s = int(math.sqrt(n0))
    pp = pos[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2)
    pp = F.interpolate(pp, scale_factor=((gh + 0.1) / math.sqrt(n0), (gw + 0.1) / math.sqrt(n0)), mode="bicubic")
    assert pp.shape[-2] == gh and pp.shape[-1] == gw
    return torch.cat((pos[:, :1], pp.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)), 1)


def normalize_state_dict(sd):
    """Map any supported checkpoint layout to a canonical dict:
    patch_w [D,C,P,P], patch_b, cls [D], pos [1,1+n,D], kind ('dinov2'|'dinov1'),
    layers: list of dict(ln1_g, ln1_b, qkv_w [3D,D], qkv_b, proj_w, proj_b, ls1|None, ln2_g, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b, ls2|None)."""
    sd = {k: v for k, v in sd.items()}
    out = {"layers": []}
    if "cls_token" in sd and "pos_embed" in sd:                      # in-repo DINO
        out["kind"] = "dinov1"
        out["patch_w"], out["patch_b"] = sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"]
        out["cls"], out["pos"] = sd["cls_token"].reshape(-1), sd["pos_embed"]
        L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        for i in range(L):
            p = f"blocks.{i}."
            out["layers"].append(dict(
                ln1_g=sd[p + "norm1.weight"], ln1_b=sd[p + "norm1.bias"], qkv_w=sd[p + "attn.qkv.weight"], qkv_b=sd[p + "attn.qkv.bias"],
                proj_w=sd[p + "attn.proj.weight"], proj_b=sd[p + "attn.proj.bias"], ls1=None,
                ln2_g=sd[p + "norm2.weight"], ln2_b=sd[p + "norm2.bias"], fc1_w=sd[p + "mlp.fc1.weight"], fc1_b=sd[p + "mlp.fc1.bias"],
                fc2_w=sd[p + "mlp.fc2.weight"], fc2_b=sd[p + "mlp.fc2.bias"], ls2=None))
        return out
    # HuggingFace: strip an optional model prefix ("dinov2." / "vit.")
    pref = ""
    for cand in ("", "dinov2.", "vit."):
        if cand + "embeddings.cls_token" in sd:
            pref = cand
            break
    g = lambda k: sd[pref + k]  # noqa: E731
    out["patch_w"], out["patch_b"] = g("embeddings.patch_embeddings.projection.weight"), g("embeddings.patch_embeddings.projection.bias")
    out["cls"], out["pos"] = g("embeddings.cls_token").reshape(-1), g("embeddings.position_embeddings")
    has_ls = any("layer_scale1" in k for k in sd)
    out["kind"] = "dinov2" if has_ls else "dinov1"
    lay = "encoder.layer." if any(k.startswith(pref + "encoder.layer.") for k in sd) else "encoder.layers."
    L = 1 + max(int(k[len(pref + lay):].split(".")[0]) for k in sd if k.startswith(pref + lay))
    for i in range(L):
        p = f"{lay}{i}."

        def first(*names):
            for n in names:
                if pref + p + n in sd:
                    return sd[pref + p + n]
            raise KeyError(f"none of {names} under {pref + p}")

        q_w, k_w, v_w = (first(f"attention.attention.{n}.weight", f"attention.{n[0]}_proj.weight") for n in ("query", "key", "value"))
        q_b, k_b, v_b = (first(f"attention.attention.{n}.bias", f"attention.{n[0]}_proj.bias") for n in ("query", "key", "value"))
        out["layers"].append(dict(
            ln1_g=first("norm1.weight", "layernorm_before.weight"), ln1_b=first("norm1.bias", "layernorm_before.bias"),
            qkv_w=torch.cat((q_w, k_w, v_w), 0), qkv_b=torch.cat((q_b, k_b, v_b), 0),
            proj_w=first("attention.output.dense.weight", "attention.o_proj.weight"), proj_b=first("attention.output.dense.bias", "attention.o_proj.bias"),
            ls1=first("layer_scale1.lambda1") if has_ls else None,
            ln2_g=first("norm2.weight", "layernorm_after.weight"), ln2_b=first("norm2.bias", "layernorm_after.bias"),
            fc1_w=first("mlp.fc1.weight", "intermediate.dense.weight"), fc1_b=first("mlp.fc1.bias", "intermediate.dense.bias"),
            fc2_w=first("mlp.fc2.weight", "output.dense.weight"), fc2_b=first("mlp.fc2.bias", "output.dense.bias"),
            ls2=first("layer_scale2.lambda1") if has_ls else None))
    return out


class ViTEngine:
    """HIP ViT forward -> last-layer key map [B, D, H/P, W/P] (f32)."""

    def __init__(self, state_dict, heads, eps=1e-6, device="cuda", full_last_layer=False, gemm_variant=0, attn_variant=0, half="f16",
                 resid="auto", ln_fold="auto"):
        """``half``: 16-bit type of the GEMM / attention operands -- "f16" (default since round 6: IEEE fp16, what the reference's fp16-autocast
        launcher multiplies in; 8x finer rounding than bf16, logits within 1e-3 of the f32 reference at full depth on the flat init; the
        configuration bench.py's headline is measured on) or "bf16" (the dtype BASELINE configs[1] names; opt-in).  Each choice is its own build of
        the same kernels (native.load).
        ``attn_variant``: 0 / 2 the product attention kernel, 1 the generic-scale kernel, 8 the fp8 path of BASELINE configs[4]; 5 / 66 =
        attn_fwd_v5_kernel / attn_fwd_v6_kernel by name (the hand-placed assembly kernels of round 4 are laboratory code: ops.attention_asm).
        ``resid``: type of the residual stream x between the GEMM epilogues and LayerNorm -- "f32" (what the reference holds), "f16"
        (IEEE fp16: half the bytes of LayerNorm's read and of the out-proj / fc2 read-modify-write) or "auto".  "auto" is a property of
        the ENGINE, never of the batch size (an image's key map does not depend on how many images travel with it):
          * bf16 operands -> fp16 stream: its 11 significand bits are 8x finer than the bf16 operands every value is rounded to before
            it is used, so it never sets the error level (logit max-abs vs the f32 oracle 3.2e-3 with either stream);
          * fp16 operands -> fp16 stream WITH LayerNorm folded into QKV / fc1 where the fold exists (D % 256 == 0, D <= 1536, a 16-bit attention
            path: ViT-B / ViT-L; logits 6.0e-4 on the flat init, the fastest configuration), f32 stream otherwise (ViT-S; 3.8e-4).
        ``ln_fold``: LayerNorm 1 / 2 of every layer but the last folded into the QKV / fc1 GEMMs (ucod_gemm_lnfold, include/ucod_dpl.h): the fp16
        stream is the GEMM's A operand, the weights carry gamma, the epilogue applies the row's (rstd, -mean * rstd).  Needs fp16 operands AND the
        fp16 stream (an MFMA takes both operands in one type); "auto" = on exactly there.
        The fp16 stream SATURATES at +-65504 and counts every saturation on the device; ``check_overflow`` (polled by every later
        ``forward``, synchronously with UCOD_CHECK_RESID=1) raises FloatingPointError when the count is non-zero: a checkpoint whose
        activations do not fit fp16 is reported instead of producing inf -> NaN key maps.
        No 16-bit configuration meets 1e-3 on trained-like weights (profiles/r06_error_budget_f16.json: every op class carries ~1e-3 of it);
        ``SplitViTEngine`` (split-operand MFMA, f32 stream: f32-equivalent) is the one that does -- the feature-cache pass uses it."""
        self.half = half
        self.lib = N.load(half)
        if resid not in ("auto", "f32", "f16"):
            raise ValueError(f"resid must be 'auto', 'f32' or 'f16', got {resid!r}")
        self.resid = resid
        c = normalize_state_dict(state_dict)
        D0 = c["patch_w"].shape[0]
        fold_shape = half == "f16" and D0 % 256 == 0 and D0 <= 1536 and attn_variant != 8       # where ucod_gemm_lnfold exists
        self.resid16 = bool(resid == "f16" or (resid == "auto" and (half == "bf16" or (fold_shape and ln_fold is not False))))
        self._ovf_host, self._ovf_events, self._ovf_dev = None, [], None
        if attn_variant not in (0, 1, 2, 8, 5, 66):
            raise ValueError(f"attn_variant must be 0, 1, 2, 5, 8 or 66 (laboratory kernels are reached through ops.attention(variant=...) / ops.attention_asm), got {attn_variant}")
        if ln_fold not in ("auto", True, False):
            raise ValueError(f"ln_fold must be 'auto', True or False, got {ln_fold!r}")
        self.kind = c["kind"]
        self.device = torch.device(device)
        self.D = c["patch_w"].shape[0]
        self.C = c["patch_w"].shape[1]
        self.P = c["patch_w"].shape[2]
        self.heads = heads
        if self.D != heads * 64:
            raise ValueError(f"head_dim must be 64 (D={self.D}, heads={heads})")
        self.L = len(c["layers"])
        self.F = c["layers"][0]["fc1_w"].shape[0]
        self.eps = float(eps)
        self.full_last_layer = bool(full_last_layer)
        self.gemm_variant, self.attn_variant = gemm_variant, attn_variant
        self.streams = 1                                           # image-parallel sub-batches on side streams (see forward)
        K = self.C * self.P * self.P
        self.Kpad = (K + 63) // 64 * 64
        dev = self.device
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()  # noqa: E731
        bf = lambda t: ops.cast_bf16(f32(t), lib=self.lib)  # noqa: E731
        pw = torch.zeros(self.D, self.Kpad, dtype=torch.float32, device=dev)
        pw[:, :K] = f32(c["patch_w"]).reshape(self.D, K)
        self._keep = []
        self._pos_src = c["pos"].detach().float().cpu()
        self._pos_cache = {}
        self.patch_w, self.patch_b, self.cls = bf(pw), f32(c["patch_b"]), f32(c["cls"])
        ones = torch.ones(self.D, dtype=torch.float32, device=dev)
        self.layers = []
        for l in c["layers"]:
            self.layers.append([f32(l["ln1_g"]), f32(l["ln1_b"]), bf(l["qkv_w"]), f32(l["qkv_b"]), bf(l["proj_w"]), f32(l["proj_b"]),
                                f32(l["ls1"]) if l["ls1"] is not None else ones, f32(l["ln2_g"]), f32(l["ln2_b"]), bf(l["fc1_w"]),
                                f32(l["fc1_b"]), bf(l["fc2_w"]), f32(l["fc2_b"]), f32(l["ls2"]) if l["ls2"] is not None else ones])
        for l in self.layers:
            l += [None, None]                                      # +14 / +15: column sums of the folded weights (ln_fold tables only)
        can_fold = fold_shape and self.resid16
        if ln_fold is True and not can_fold:
            raise ValueError("ln_fold needs half='f16' with the fp16 residual stream (resid='f16' or 'auto'), D % 256 == 0 and a 16-bit attention path")
        self.ln_fold = bool(can_fold and ln_fold is not False)
        self.fold_layers = None
        if self.ln_fold:
            self.fold_layers = []
            for l, src in zip(self.layers, c["layers"]):
                # (the folded QKV entries also carry the softmax pre-scale head_dim^-0.5 * log2 e on their Q rows -- what the unfolded pass applies as
                # a column scale in the epilogue -- so that the folded epilogue has no per-column multiply left; attn_variant 1 takes Q unscaled)
                # (set-up arithmetic on the HOST in f64 -- fold.py -- then one copy per tensor: no vendor-BLAS kernel runs on the device for it)
                cpu = lambda t: t.detach().to("cpu", torch.float32)  # noqa: E731
                qs = torch.ones(3 * self.D, dtype=torch.float32)
                if attn_variant != 1:
                    qs[:self.D] = 0.125 * 1.4426950408889634
                qw, qb, qc = (t.to(dev) for t in self._fold(cpu(src["ln1_g"]), cpu(src["ln1_b"]), cpu(src["qkv_w"]), cpu(src["qkv_b"]), row_scale=qs))
                fw, fb, fc = (t.to(dev) for t in self._fold(cpu(src["ln2_g"]), cpu(src["ln2_b"]), cpu(src["fc1_w"]), cpu(src["fc1_b"])))
                fl = list(l)
                fl[2], fl[3], fl[9], fl[10], fl[14], fl[15] = qw, qb, fw, fb, qc, fc
                self.fold_layers.append(fl)
        self._ws = None
        self._ws_key = None

    def _fold(self, gamma, beta, w, b, row_scale=None):
        """LayerNorm(gamma, beta) followed by Linear(w, b) [* row_scale per output] as one GEMM on the un-normalised rows (include/ucod_dpl.h:
        ucod_gemm_lnfold): W' = 16-bit(q (.) W (.) gamma), column sums of the ROUNDED W' (what the MFMA multiplies by), b' = q (W beta + b) in f64."""
        from .fold import fold_layernorm_linear
        return fold_layernorm_linear(gamma, beta, w, b, row_scale=row_scale, half=torch.float16)      # (the fold exists in the fp16-operand build only)

    def _table(self, gh, gw, L=None):
        """(ctypes table, tensors kept alive) of a pass over the first L layers.  With ln_fold every layer but the last OF THE PASS contributes its
        folded entries; the last keeps the plain ones (its LayerNorm 1 runs as a kernel and feeds the key hook)."""
        L = self.L if L is None else L
        ptrs = [self.patch_w, self.patch_b, self.cls, self._pos(gh, gw)]
        for i, l in enumerate(self.layers):
            ptrs += self.fold_layers[i] if (self.ln_fold and i != L - 1) else l
        return (C.c_void_p * len(ptrs))(*[None if t is None else t.data_ptr() for t in ptrs]), ptrs

    def param_bytes(self):
        n = self.patch_w.numel() * 2
        for l in self.layers:
            n += sum(t.numel() * t.element_size() for t in l if t is not None)
        # the folded tables are a SECOND copy of every layer's QKV / fc1 weights (+ their folded biases and column sums); the plain copies stay because
        # the last layer of a (possibly truncated) pass runs unfolded
        for fl, l in zip(self.fold_layers or (), self.layers):
            n += sum(t.numel() * t.element_size() for t, t0 in zip(fl, l) if t is not None and t is not t0)
        return n

    def _pos(self, gh, gw):
        key = (gh, gw)
        if key not in self._pos_cache:
            fn = _interp_pos_dinov2 if self.kind == "dinov2" else _interp_pos_dinov1
            self._pos_cache[key] = fn(self._pos_src, gh, gw)[0].to(self.device, torch.float32).contiguous()
        return self._pos_cache[key]

    def _desc(self, B, H, W, L=None):
        d = N.VitDesc()
        d.B, d.C, d.H, d.W, d.P = B, self.C, H, W, self.P
        d.D, d.heads, d.F, d.L, d.Kpad = self.D, self.heads, self.F, (self.L if L is None else L), self.Kpad
        d.eps = self.eps
        d.full_last_layer = int(self.full_last_layer)
        d.gemm_variant, d.attn_variant = self.gemm_variant, self.attn_variant
        d.resid16 = int(self.resid16)
        d.ln_fold = int(getattr(self, "ln_fold", False))
        return d

    # ---- fp16 residual stream: saturation guard --------------------------------------------------------------------------------
    def _own_counter(self):
        """Context: the library's fp16-stream kernels launched from this thread count into THIS engine's device word (ucod_resid16_overflow_bind),
        so that one engine's saturation is never reported by another engine that shares the GPU."""
        eng = self

        class _Bound:
            def __enter__(self_):
                if eng._ovf_dev is None:
                    eng._ovf_dev = torch.zeros(1, dtype=torch.int32, device=eng.device)
                N.check(eng.lib.ucod_resid16_overflow_bind(eng._ovf_dev.data_ptr()), "ucod_resid16_overflow_bind")

            def __exit__(self_, *exc):
                eng.lib.ucod_resid16_overflow_bind(None)
                return False
        return _Bound()

    def _arm_overflow_check(self, stream, used_resid16=None):
        """Enqueue an asynchronous copy of the device's saturation counter behind the pass just launched on ``stream``.  ``used_resid16``:
        whether THAT pass ran the fp16 stream (default: the engine's setting; the LoRA engine's no-grad teacher pass says so itself)."""
        if not (self.resid16 if used_resid16 is None else used_resid16):
            return
        if self._ovf_host is None:
            self._ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        with self._own_counter():
            N.check(self.lib.ucod_resid16_overflow_fetch(self._ovf_host.data_ptr(), stream.cuda_stream), "ucod_resid16_overflow_fetch")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._ovf_events.append(ev)
        del self._ovf_events[:-8]                                # (only the latest copies matter: the counter never decreases)

    def check_overflow(self, wait=False):
        """Raise FloatingPointError if any finished pass saturated the fp16 residual stream.  ``wait``: block until every pass enqueued so far
        has finished (otherwise only passes that already have)."""
        if self._ovf_host is None:                              # no pass of this engine has used the fp16 stream
            return
        if wait:
            for ev in self._ovf_events:
                ev.synchronize()
            self._ovf_events = []
        elif not any(ev.query() for ev in self._ovf_events):
            return
        n = int(self._ovf_host[0])
        if n > 0:
            # before the reset: every fetch still queued on a side stream must have landed, or its (stale, non-zero) count would arrive
            # AFTER the host word is cleared and raise a second time for the same saturation
            for ev in self._ovf_events:
                ev.synchronize()
            self._ovf_events = []
            with self._own_counter():
                N.check(self.lib.ucod_resid16_overflow_reset(N.stream()), "ucod_resid16_overflow_reset")
            torch.cuda.current_stream(self.device).synchronize()
            self._ovf_host.zero_()
            how = "build the engine with resid='f32'" if self.resid16 else "call forward_nograd(..., resid16=False)"
            raise FloatingPointError(f"the fp16 residual stream of this engine saturated at +-65504 (or met a NaN) in {n} wave-lane(s)"
                                     + (" -- or a folded LayerNorm met a row with |mean| > 256 sigma (outside the fold's range)" if getattr(self, "ln_fold", False) else "")
                                     + f": the activations do not fit this configuration; {how}.  The counter is polled without blocking: the pass (or, for a training engine, the "
                                     f"optimiser step or steps) that consumed the clamped activations has already been applied -- discard its results "
                                     f"(UCOD_CHECK_RESID=1 checks synchronously after every pass).")

    _sync_check = os.environ.get("UCOD_CHECK_RESID") == "1"    # debug: check the saturation counter synchronously after every pass

    def forward(self, img, out=None, _async=False, n_layers=None):
        """``n_layers``: stop after that many encoder layers and return THAT layer's key map (diagnostics: the per-layer error
        table of bench.py); default = the whole backbone."""
        if n_layers is not None and not 1 <= n_layers <= self.L:
            raise ValueError(f"n_layers must be in [1, {self.L}]")
        if not img.is_cuda:
            raise RuntimeError("ViTEngine needs a CUDA(ROCm) tensor; there is no CPU path")
        img = img.to(torch.float32).contiguous()
        B, Cc, H, W = img.shape
        gh, gw = H // self.P, W // self.P
        lib = self.lib
        table, _keep = self._table(gh, gw, n_layers)
        key = out if out is not None else torch.empty(B, self.D, gh, gw, dtype=torch.float32, device=self.device)
        self.check_overflow()                                      # (non-blocking) passes that have finished since the last call
        ns = max(1, min(int(getattr(self, "streams", 1)), B))
        if ns == 1 and not _async:
            d = self._desc(B, H, W, n_layers)
            need = lib.ucod_vit_workspace_bytes(C.byref(d))
            if need == 0:
                raise ValueError("unsupported ViT geometry")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            with self._own_counter():
                N.check(lib.ucod_vit_forward(C.byref(d), table, N.ptr(img), N.ptr(key), N.ptr(self._ws), self._ws.numel(), N.stream()),
                        "ucod_vit_forward")
            self._arm_overflow_check(torch.cuda.current_stream(self.device))
            if self._sync_check:
                self.check_overflow(wait=True)
            return key
        # Image-parallel sub-batches on independent HIP streams: every kernel of the pass is per-image, so the halves are
        # independent and the tail of one stream's GEMM (a partial last round of 256-row tiles leaves most CUs idle) is filled by
        # the other stream's next kernel.  Results are identical to the single-stream pass (same kernels, same per-image math).
        if getattr(self, "_side", None) is None or len(self._side) != ns:
            # Normal priority.  High-priority side streams (UCOD_SIDE_PRIORITY=-1) were measured: nothing on the pipelined frozen-backbone
            # step (10.16 vs 10.19 ms) and MINUS 17 % in backbone-backward mode (585 vs 702 images/s: the EMA teacher's pass on these
            # streams then starves the student's forward / backward on the training streams).
            prio = int(os.environ.get("UCOD_SIDE_PRIORITY", "0"))
            self._side = [torch.cuda.Stream(device=self.device, priority=prio) for _ in range(ns)]
            self._side_ws = [None] * ns
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        bounds = [B * i // ns for i in range(ns + 1)]
        events = []
        for i in range(ns):
            b0, b1 = bounds[i], bounds[i + 1]
            d = self._desc(b1 - b0, H, W, n_layers)
            need = lib.ucod_vit_workspace_bytes(C.byref(d))
            if need == 0:
                raise ValueError("unsupported ViT geometry")
            if self._side_ws[i] is None or self._side_ws[i].numel() < need:
                self._side_ws[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
            st = self._side[i]
            st.wait_event(start)
            img.record_stream(st)                                  # allocated on the caller's stream, consumed on this one: keep the
            key.record_stream(st)                                  # caching allocator from recycling them before this stream is done
            with torch.cuda.stream(st):
                with self._own_counter():
                    N.check(lib.ucod_vit_forward(C.byref(d), table, N.ptr(img[b0:b1]), N.ptr(key[b0:b1]), N.ptr(self._side_ws[i]),
                                                 self._side_ws[i].numel(), N.stream()), "ucod_vit_forward")
                self._arm_overflow_check(st)
                done = torch.cuda.Event()
                done.record(st)
            events.append(done)
        if _async:
            return key, events
        for done in events:
            cur.wait_event(done)
        if self._sync_check:
            self.check_overflow(wait=True)
        return key

    def forward_with_cls_attention(self, img):
        """(key [B,D,h,w], att [B,heads,h*w]): the key map plus ``outputs.attentions[-1][:, :, 0, 1:]`` of the HF model -- the CLS
        query's softmax row of the LAST layer, which is all the pseudo-label generator reads from the attentions
        (data/utils/found_bkg_mask.py:23; generate_pseudo_label.py:78-89).  Single-stream pass: the CLS projections are taken
        from the LayerNorm-1 output the pass leaves in its workspace."""
        if self.full_last_layer:
            raise RuntimeError("forward_with_cls_attention needs the key-minimal pass (full_last_layer=False)")
        ns, self.streams = self.streams, 1
        try:
            key = self.forward(img)
        finally:
            self.streams = ns
        B, _, H, W = img.shape
        lib = self.lib
        d = self._desc(B, H, W)
        off = lib.ucod_vit_last_ln1_offset(C.byref(d))
        tok = key.shape[-2] * key.shape[-1] + 1
        last = self.layers[-1]
        q = torch.empty(B, self.D, dtype=torch.float32, device=self.device)
        k = torch.empty(B, self.D, dtype=torch.float32, device=self.device)
        N.check(lib.ucod_cls_qk(self._ws.data_ptr() + off, N.ptr(last[2]), N.ptr(last[3]), N.ptr(q), N.ptr(k), B, tok, self.D, N.stream()), "ucod_cls_qk")
        att = torch.empty(B, self.heads, tok - 1, dtype=torch.float32, device=self.device)
        N.check(lib.ucod_cls_attention(N.ptr(q), N.ptr(k), N.ptr(key), N.ptr(att), B, self.heads, tok - 1, 0.125, N.stream()), "ucod_cls_attention")
        return key, att

    def forward_async(self, img, out=None):
        """Enqueue the pass on the side streams WITHOUT making the current stream wait: returns (key, events); the consumer
        calls ``torch.cuda.current_stream().wait_event(e)`` for each event before reading ``key``.  The side streams start
        after everything already enqueued on the current stream (so a ring of key buffers is safe to reuse)."""
        return self.forward(img, out=out, _async=True)

    __call__ = forward


class SplitViTEngine:
    """The frozen backbone at the REFERENCE's cached-feature precision: HIP ViT forward -> last-layer key map [B, D, H/P, W/P] (f32) with every matrix product
    computed on split operands (include/ucod_dpl.h: "split-operand backbone pass"; csrc/split.hip) -- f32 residual stream, f32 LayerNorm, exact-erf GELU,
    f32 softmax, each f32 GEMM / attention operand as a sum of ``terms`` bf16 values on the bf16 MFMA:

    * ``terms=2``: 16 significand bits per operand, 3x the matrix work of the 16-bit engines; mask logits within ~3e-5 of the f32 oracle at full size on the
      trained-like weights (tests/test_gpu_split.py), i.e. the configuration that meets the 1e-3 bar where the 16-bit engines do not;
    * ``terms=3`` (default): 24 bits, f32-equivalent, 6x -- what ``build_feature_cache`` uses by default, because the reference runs that pass in plain fp32
      (/root/reference/data/datasets/base_dataset.py:124-138: no autocast) and its speed does not matter.

    Same call interface as ``ViTEngine`` (``forward`` / ``forward_async`` / ``__call__``, ``D``, ``P``, ``streams``); key-minimal pass only."""

    def __init__(self, state_dict, heads, eps=1e-6, device="cuda", terms=3, gemm_variant=0):
        if terms not in (2, 3):
            raise ValueError(f"terms must be 2 or 3, got {terms!r}")
        self.terms, self.half = int(terms), f"bf16x{int(terms)}"
        self.lib = N.load("bf16")                                 # (bf16 MFMA; the fp16 build refuses the split entry points)
        self.nprod = ops.split_products(terms)
        c = normalize_state_dict(state_dict)
        self.kind, self.device = c["kind"], torch.device(device)
        self.D, self.C, self.P = c["patch_w"].shape[0], c["patch_w"].shape[1], c["patch_w"].shape[2]
        self.heads = heads
        if self.D != heads * 64:
            raise ValueError(f"head_dim must be 64 (D={self.D}, heads={heads})")
        self.L, self.F, self.eps = len(c["layers"]), c["layers"][0]["fc1_w"].shape[0], float(eps)
        self.gemm_variant, self.streams = gemm_variant, 1
        self.resid16 = self.ln_fold = self.full_last_layer = False
        self.resid, self.attn_variant = "f32", 0
        K = self.C * self.P * self.P
        self.Kpad = (K + 63) // 64 * 64
        dev = self.device
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()  # noqa: E731
        sw = lambda t, role=1: ops.split_rows(f32(t), self.terms, role)  # noqa: E731      (weights: the B side of y = x W^T)
        pw = torch.zeros(self.D, self.Kpad, dtype=torch.float32, device=dev)
        pw[:, :K] = f32(c["patch_w"]).reshape(self.D, K)
        self._pos_src, self._pos_cache = c["pos"].detach().float().cpu(), {}
        self.patch_w, self.patch_b, self.cls = sw(pw), f32(c["patch_b"]), f32(c["cls"])
        ones = torch.ones(self.D, dtype=torch.float32, device=dev)
        self.layers = []
        D = self.D
        for l in c["layers"]:
            self.layers.append([f32(l["ln1_g"]), f32(l["ln1_b"]), sw(l["qkv_w"]), f32(l["qkv_b"]), sw(l["proj_w"]), f32(l["proj_b"]),
                                f32(l["ls1"]) if l["ls1"] is not None else ones, f32(l["ln2_g"]), f32(l["ln2_b"]), sw(l["fc1_w"]),
                                f32(l["fc1_b"]), sw(l["fc2_w"]), f32(l["fc2_b"]), f32(l["ls2"]) if l["ls2"] is not None else ones,
                                sw(l["qkv_w"][D:2 * D], role=0), None])      # +14: the K rows as the A side of the key hook's GEMM
        last = c["layers"][-1]                                   # f32 copies for forward_with_cls_attention: the CLS query / key of the LAST layer
        self._last = dict(ln_g=f32(last["ln1_g"]), ln_b=f32(last["ln1_b"]), wq=f32(last["qkv_w"][:D]), bq=f32(last["qkv_b"][:D]),
                          wk=f32(last["qkv_w"][D:2 * D]), bk=f32(last["qkv_b"][D:2 * D]))
        self._ws = None
        self._side = self._side_ws = None

    _pos = ViTEngine._pos
    _desc = ViTEngine._desc

    def param_bytes(self):
        return self.patch_w.numel() * 2 + sum(t.numel() * t.element_size() for l in self.layers for t in l if t is not None)

    def check_overflow(self, wait=False):
        """(interface of ViTEngine: the f32 stream has nothing to saturate)"""
        return None

    def _table(self, gh, gw):
        ptrs = [self.patch_w, self.patch_b, self.cls, self._pos(gh, gw)]
        for l in self.layers:
            ptrs += l
        return (C.c_void_p * len(ptrs))(*[None if t is None else t.data_ptr() for t in ptrs]), ptrs

    def _run(self, img, key, ws_slot, n_layers=None):
        B, _, H, W = img.shape
        d = self._desc(B, H, W, n_layers)
        d.resid16 = d.ln_fold = d.full_last_layer = d.attn_variant = 0
        need = self.lib.ucod_vit_split_workspace_bytes(C.byref(d), self.terms)
        if need == 0:
            raise ValueError("unsupported ViT geometry")
        ws = ws_slot[0]
        if ws is None or ws.numel() < need:
            ws = ws_slot[0] = torch.empty(need, dtype=torch.uint8, device=self.device)
        table, _keep = self._table(H // self.P, W // self.P)
        N.check(self.lib.ucod_vit_forward_split(C.byref(d), self.terms, table, N.ptr(img), N.ptr(key), N.ptr(ws), ws.numel(), N.stream()), "ucod_vit_forward_split")

    def forward(self, img, out=None, _async=False, n_layers=None):
        if n_layers is not None and not 1 <= n_layers <= self.L:
            raise ValueError(f"n_layers must be in [1, {self.L}]")
        if not img.is_cuda:
            raise RuntimeError("SplitViTEngine needs a CUDA(ROCm) tensor; there is no CPU path")
        img = img.to(torch.float32).contiguous()
        B, _, H, W = img.shape
        key = out if out is not None else torch.empty(B, self.D, H // self.P, W // self.P, dtype=torch.float32, device=self.device)
        if not _async:
            if self._ws is None:
                self._ws = [None]
            self._run(img, key, self._ws, n_layers)
            return key
        # forward_async: the whole pass on ONE side stream (its workspace is several GB at batch 32: no image-parallel halves here)
        if self._side is None:
            self._side, self._side_ws = torch.cuda.Stream(device=self.device), [None]
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        self._side.wait_event(start)
        img.record_stream(self._side)
        key.record_stream(self._side)
        with torch.cuda.stream(self._side):
            self._run(img, key, self._side_ws, n_layers)
            done = torch.cuda.Event()
            done.record(self._side)
        return key, [done]

    def forward_async(self, img, out=None):
        return self.forward(img, out=out, _async=True)

    def forward_with_cls_attention(self, img):
        """(key [B,D,h,w], att [B,heads,h*w]) like ``ViTEngine.forward_with_cls_attention``: the key map plus ``outputs.attentions[-1][:, :, 0, 1:]`` of the HF model
        (data/utils/found_bkg_mask.py:23; generate_pseudo_label.py:78-89 -- a plain fp32 pass under torch.no_grad() in the reference, which is why the
        pseudo-label generator asks for this engine).  The CLS rows of the last layer's input are taken from the f32 residual stream the pass leaves in its
        workspace, normalised by the f32 LayerNorm kernel, projected to the CLS query / key on split operands, and fed to the f32 attention-row kernel."""
        key = self.forward(img)
        B, _, H, W = img.shape
        d = self._desc(B, H, W)
        d.resid16 = d.ln_fold = d.full_last_layer = d.attn_variant = 0
        off = self.lib.ucod_vit_split_stream_offset(C.byref(d), self.terms)
        tok = key.shape[-2] * key.shape[-1] + 1
        x = self._ws[0][off:off + B * tok * self.D * 4].view(torch.float32).view(B, tok, self.D)
        L_ = self._last
        h_cls = ops.layernorm(x[:, 0].contiguous(), L_["ln_g"], L_["ln_b"], self.eps, out_f32=True)
        q = ops.linear_split(h_cls, L_["wq"], L_["bq"], self.terms)
        k = ops.linear_split(h_cls, L_["wk"], L_["bk"], self.terms)
        att = torch.empty(B, self.heads, tok - 1, dtype=torch.float32, device=self.device)
        N.check(self.lib.ucod_cls_attention(N.ptr(q), N.ptr(k), N.ptr(key), N.ptr(att), B, self.heads, tok - 1, 0.125, N.stream()), "ucod_cls_attention")
        return key, att

    __call__ = forward


_QKV = ("query", "key", "value")


class ViTLoRAEngine(ViTEngine):

    """Backbone-backward mode: the frozen ViT with peft-style LoRA on query / key / value of every
    encoder layer, as models/modules/full_model.py:47-72 configures it (r=2, lora_alpha=4, bias='none', target
    query/key/value; lora_A kaiming_uniform(a=sqrt 5), lora_B zeros).  ``forward_train`` saves activations,
    ``backward`` turns the cotangent of the key map into LoRA gradients -- both entirely in the HIP library
    (``ucod_vit_forward_train`` / ``ucod_vit_backward``).  LoRA dropout (``lora_dropout``, 0.05 in the reference config) is a counter-based
    hash mask on the LoRA branch's input in ``train()`` mode and off in ``eval()`` (include/ucod_dpl.h: ucod_lora_dropout).

    Parameters live in ONE flat f32 arena ``self.lora`` [L, 6*r*D] (layer-major: A_q | B_q | A_k | B_k | A_v | B_v), gradients
    in ``self.lora_grad`` with the same layout -- ready for a single flat all-reduce and the fused AdamW kernel."""

    def __init__(self, state_dict, heads, r=2, lora_alpha=4, eps=1e-6, device="cuda", gemm_variant=0, generator=None, lora_dropout=0.0,
                 seed=0, resid="auto"):
        # resid: as ViTEngine ("auto" = the fp16 residual stream with bf16 operands).  Round 4: the training pass SAVES the stream in that
        # type and LayerNorm backward reads it (ucod_layernorm_bwd_ex); resid="f32" keeps the round-3 path.
        super().__init__(state_dict, heads, eps=eps, device=device, full_last_layer=False, gemm_variant=gemm_variant, attn_variant=2, resid=resid,
                         half="bf16")                             # (the backbone-backward entry points exist in the bf16 build only: include/ucod_dpl.h, ucod_half_name)
        if not 0.0 <= lora_dropout < 1.0:
            raise ValueError("lora_dropout must be in [0, 1)")
        if 0.0 < lora_dropout < 1.0 / 1024:
            raise ValueError("lora_dropout below 1/1024 would be rounded to no dropout at all (the mask threshold is floor(1024 p): include/ucod_dpl.h, "
                             "ucod_lora_dropout); use 0 or a value >= 1/1024")
        # LoRA dropout (LoraConfig.lora_dropout, full_model.py:50): active while `self.training` is True; the mask of a step is a pure
        # function of (seed, step), regenerated by the backward kernels.  `eval()` / `train()` switch it like nn.Module does.
        self.lora_dropout, self.training = float(lora_dropout), True
        self._seed, self._step, self._step_seed = int(seed), 0, 0
        if r < 1 or 3 * r > N.LORA_AUG:
            raise ValueError(f"LoRA rank {r} unsupported (1 <= r <= {N.LORA_AUG // 3})")
        self.r, self.scaling = int(r), float(lora_alpha) / float(r)
        D, L, dev = self.D, self.L, self.device
        self.lora = torch.zeros(L, 6 * r * D, dtype=torch.float32, device=dev)
        bound = 1.0 / math.sqrt(D)                      # kaiming_uniform_(a=sqrt(5)) on [r, D]: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        a = (torch.rand(L, 3, r * D, generator=generator) * 2 - 1) * bound
        for p in range(3):
            self.lora[:, p * 2 * r * D:p * 2 * r * D + r * D] = a[:, p].to(dev)
        self.lora_grad = torch.zeros_like(self.lora)
        A = N.LORA_AUG
        self.train_layers = []
        for l in self.layers:
            qkv_w, proj_w, fc1_w, fc2_w = l[2], l[4], l[9], l[11]
            w_aug = torch.zeros(3 * D, D + A, dtype=torch.bfloat16, device=dev)
            w_aug[:, :D] = qkv_w
            wt_aug = torch.zeros(D, 3 * D + A, dtype=torch.bfloat16, device=dev)
            wt_aug[:, :3 * D] = qkv_w.t()
            self.train_layers.append([w_aug, wt_aug, proj_w.t().contiguous(), fc1_w.t().contiguous(), fc2_w.t().contiguous()])
        self._tside = None
        self._saved_for = None
        self.repack()

    # ---- parameters -------------------------------------------------------------------------------------------------
    def _slices(self, p):
        rD = self.r * self.D
        return slice(p * 2 * rD, p * 2 * rD + rD), slice(p * 2 * rD + rD, (p + 1) * 2 * rD)

    def lora_state_dict(self, grads=False, prefix="encoder.layer."):
        """peft-style names: encoder.layer.{i}.attention.attention.{query,key,value}.lora_{A,B}.weight"""
        src = self.lora_grad if grads else self.lora
        out = {}
        for i in range(self.L):
            for p, name in enumerate(_QKV):
                sa, sb = self._slices(p)
                base = f"{prefix}{i}.attention.attention.{name}."
                out[base + "lora_A.weight"] = src[i, sa].reshape(self.r, self.D).clone()
                out[base + "lora_B.weight"] = src[i, sb].reshape(self.D, self.r).clone()
        return out

    def load_lora_state_dict(self, sd, prefix="encoder.layer."):
        for i in range(self.L):
            for p, name in enumerate(_QKV):
                sa, sb = self._slices(p)
                base = f"{prefix}{i}.attention.attention.{name}."
                self.lora[i, sa] = sd[base + "lora_A.weight"].to(self.device, torch.float32).reshape(-1)
                self.lora[i, sb] = sd[base + "lora_B.weight"].to(self.device, torch.float32).reshape(-1)
        self.repack()

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _drop_p(self):
        return self.lora_dropout if self.training else 0.0

    def repack(self):
        """Refresh the LoRA columns of the augmented weights (call after every parameter update).  With dropout active the A^T
        columns of the dgrad weight are zero: the masked t A term is added by the LayerNorm-1 backward instead."""
        lib = N.load()
        zero_a = int(self._drop_p() > 0.0)
        for i, tl in enumerate(self.train_layers):
            N.check(lib.ucod_lora_pack(N.ptr(self.lora[i]), self.r, self.scaling, N.ptr(tl[0]), N.ptr(tl[1]), self.D, zero_a, N.stream()), "ucod_lora_pack")
        self._packed_zero_a = zero_a

    # ---- passes -----------------------------------------------------------------------------------------------------
    def _train_desc(self, B, H, W):
        t = N.VitTrainDesc()
        t.vit = self._desc(B, H, W)
        t.lora_r, t.lora_scaling = self.r, self.scaling
        t.lora_dropout, t.seed = self._drop_p(), self._step_seed
        return t

    def _tables(self, gh, gw, grad=None):
        pos = self._pos(gh, gw)
        ptrs = [self.patch_w, self.patch_b, self.cls, pos]
        for l in self.layers:
            ptrs += l
        grad = self.lora_grad if grad is None else grad
        tptrs = []
        for i, tl in enumerate(self.train_layers):
            tptrs += tl + [self.lora[i], grad[i]]
        keep = ptrs + tptrs
        return (C.c_void_p * len(ptrs))(*[None if t is None else t.data_ptr() for t in ptrs]), (C.c_void_p * len(tptrs))(*[t.data_ptr() for t in tptrs]), keep

    def _chunks(self, B):
        """Image-parallel sub-batches (``self.train_streams``, default 2): as in ViTEngine.forward, every kernel of both passes is
        per image except the LoRA-gradient reduction over rows, which is done per chunk into its own buffer and summed at the end."""
        ns = max(1, min(int(getattr(self, "train_streams", 2)), B))
        if getattr(self, "_tside", None) is None or len(self._tside) != ns:
            self._tside = [torch.cuda.Stream(device=self.device) for _ in range(ns)]
            self._tside_ws = [None] * ns
            self._tside_grad = [torch.zeros_like(self.lora) for _ in range(ns)]
            self._chunk_seed = [0] * ns
        return [(B * i // ns, B * (i + 1) // ns) for i in range(ns)]

    def _fan_out(self, fn, tensors=()):
        """Run fn(i, b0, b1) for every chunk on its side stream, between two events on the current stream.  ``tensors``: caller-stream
        tensors the side streams touch (recorded on them so the caching allocator does not recycle their memory early)."""
        cur = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(cur)
        for i, (b0, b1) in enumerate(self._bounds):
            st = self._tside[i]
            st.wait_event(start)
            for t in tensors:
                t.record_stream(st)
            with torch.cuda.stream(st):
                fn(i, b0, b1)
                done = torch.cuda.Event()
                done.record(st)
            cur.wait_event(done)

    def forward_train(self, img, out=None):
        if not img.is_cuda:
            raise RuntimeError("ViTLoRAEngine needs a CUDA(ROCm) tensor; there is no CPU path")
        img = img.to(torch.float32).contiguous()
        B, _, H, W = img.shape
        gh, gw = H // self.P, W // self.P
        lib = N.load()
        key = out if out is not None else torch.empty(B, self.D, gh, gw, dtype=torch.float32, device=self.device)
        self.check_overflow()                                      # (non-blocking) training passes on the fp16 stream that have finished since the last call
        self._bounds = self._chunks(B)
        self._step += 1
        self._step_seed = (self._seed * 0x9E3779B97F4A7C15 + self._step) & 0xFFFFFFFFFFFFFFFF     # fresh masks every step
        if int(self._drop_p() > 0.0) != getattr(self, "_packed_zero_a", 0):
            self.repack()                                           # train()/eval() changed which side carries the A term

        def run(i, b0, b1):
            t = self._train_desc(b1 - b0, H, W)
            t.seed = (self._step_seed + 0x51ED270B * (i + 1) * int(b0 > 0)) & 0xFFFFFFFFFFFFFFFF   # (chunks index rows from 0: decorrelate them)
            self._chunk_seed[i] = t.seed
            need = lib.ucod_vit_train_workspace_bytes(C.byref(t))
            if need == 0:
                raise ValueError("unsupported ViT geometry")
            if self._tside_ws[i] is None or self._tside_ws[i].numel() < need:
                self._tside_ws[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
            T, TT, keep = self._tables(gh, gw, self._tside_grad[i])
            with self._own_counter():
                N.check(lib.ucod_vit_forward_train(C.byref(t), T, TT, N.ptr(img[b0:b1]), N.ptr(key[b0:b1]), N.ptr(self._tside_ws[i]),
                                                   self._tside_ws[i].numel(), N.stream()), "ucod_vit_forward_train")
            self._arm_overflow_check(torch.cuda.current_stream(self.device))     # (no-op with the f32 stream)

        self._fan_out(run, (img, key))
        self._saved_for = (B, H, W)
        return key

    def forward_nograd(self, img, out=None, resid16=True):
        """The pass of a LoRA backbone that needs no backward -- the EMA teacher (models/modules/full_model.py:84,108-111: backbone_ema under
        torch.no_grad()): same arithmetic as ``forward_train`` (LoRA aug columns, dropout masks of this engine's own seed stream when in
        train mode) with nothing saved, an inference-sized workspace and, by default, the fp16 residual stream of the frozen-backbone pass
        (bf16 operands: the stream's rounding is 8x finer than theirs).  Image-parallel chunks on the training side streams."""
        if not img.is_cuda:
            raise RuntimeError("ViTLoRAEngine needs a CUDA(ROCm) tensor; there is no CPU path")
        img = img.to(torch.float32).contiguous()
        B, _, H, W = img.shape
        gh, gw = H // self.P, W // self.P
        lib = N.load()
        key = out if out is not None else torch.empty(B, self.D, gh, gw, dtype=torch.float32, device=self.device)
        self._bounds = self._chunks(B)
        self._step += 1
        self._step_seed = (self._seed * 0x9E3779B97F4A7C15 + self._step) & 0xFFFFFFFFFFFFFFFF
        if int(self._drop_p() > 0.0) != getattr(self, "_packed_zero_a", 0):
            self.repack()
        if getattr(self, "_iside_ws", None) is None or len(self._iside_ws) != len(self._tside):
            self._iside_ws = [None] * len(self._tside)

        self.check_overflow()                                      # (non-blocking) teacher passes that have finished since the last call

        def run(i, b0, b1):
            t = self._train_desc(b1 - b0, H, W)
            t.vit.resid16 = int(bool(resid16))
            t.seed = (self._step_seed + 0x51ED270B * (i + 1) * int(b0 > 0)) & 0xFFFFFFFFFFFFFFFF
            need = lib.ucod_vit_lora_infer_workspace_bytes(C.byref(t))
            if need == 0:
                raise ValueError("unsupported ViT geometry")
            if self._iside_ws[i] is None or self._iside_ws[i].numel() < need:
                self._iside_ws[i] = torch.empty(need, dtype=torch.uint8, device=self.device)
            T, TT, keep = self._tables(gh, gw, self._tside_grad[i])
            with self._own_counter():
                N.check(lib.ucod_vit_forward_lora_infer(C.byref(t), T, TT, N.ptr(img[b0:b1]), N.ptr(key[b0:b1]), N.ptr(self._iside_ws[i]),
                                                        self._iside_ws[i].numel(), N.stream()), "ucod_vit_forward_lora_infer")
            # this engine's own stream type is f32 (its backward reads it), but THIS pass may run the fp16 one: its saturation counter is
            # fetched behind every chunk and polled by the next call / check_overflow(wait=True) at the loop's boundaries
            self._arm_overflow_check(torch.cuda.current_stream(self.device), used_resid16=bool(resid16))

        self._fan_out(run, (img, key))
        return key

    def backward(self, dkey):
        """dkey [B, D, H/P, W/P] -> self.lora_grad (overwritten), returned as the flat [L, 6*r*D] tensor."""
        if self._saved_for is None:
            raise RuntimeError("backward() without a preceding forward_train()")
        B, H, W = self._saved_for
        gh, gw = H // self.P, W // self.P
        dkey = dkey.to(torch.float32).contiguous()
        if tuple(dkey.shape) != (B, self.D, gh, gw):
            raise ValueError(f"dkey shape {tuple(dkey.shape)} != {(B, self.D, gh, gw)}")
        lib = N.load()

        def run(i, b0, b1):
            t = self._train_desc(b1 - b0, H, W)
            t.seed = self._chunk_seed[i]
            T, TT, keep = self._tables(gh, gw, self._tside_grad[i])
            N.check(lib.ucod_vit_backward(C.byref(t), T, TT, N.ptr(dkey[b0:b1]), N.ptr(self._tside_ws[i]), self._tside_ws[i].numel(), N.stream()),
                    "ucod_vit_backward")

        self._fan_out(run, (dkey,))
        torch.sum(torch.stack(self._tside_grad[:len(self._bounds)]), dim=0, out=self.lora_grad) if len(self._bounds) > 1 else self.lora_grad.copy_(self._tside_grad[0])
        self._saved_for = None
        return self.lora_grad

    # ---- torch.autograd / nn.Module integration ----------------------------------------------------------------------
    def clone_for_ema(self):
        """A second engine over the SAME frozen weight tensors with its own LoRA arena (models/modules/full_model.py:84:
        ``backbone_ema = copy.deepcopy(backbone)``; only the LoRA matrices can ever differ, so only they are duplicated)."""
        other = object.__new__(ViTLoRAEngine)
        other.__dict__.update(self.__dict__)
        other.lora = self.lora.clone()
        other.lora_grad = torch.zeros_like(self.lora)
        other.train_layers = [[tl[0].clone(), tl[1].clone()] + tl[2:] for tl in self.train_layers]
        other._tside, other._saved_for = None, None
        other._pos_cache = dict(self._pos_cache)
        # The reference's teacher is a deepcopy that stays in train mode (full_model.py:84; nobody calls eval() on it), so its LoRA
        # dropout is active too, with masks drawn independently of the student's: own seed stream, own step counter.
        other._seed = (self._seed * 0x2545F4914F6CDD1D + 0x5DEECE66D) & 0xFFFFFFFFFFFFFFFF
        other._step, other._step_seed = 0, 0
        other.repack()
        return other

    def apply(self, img, lora_param):
        """Differentiable call: key = f(img; lora_param).  ``lora_param`` must be a tensor sharing storage with ``self.lora``
        (see ``LoRABackbone``); its ``.grad`` receives ``self.lora_grad`` on backward."""
        return _LoRABackboneFn.apply(img, lora_param, self)


class _LoRABackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, lora_param, engine):
        if lora_param.data_ptr() != engine.lora.data_ptr():
            raise RuntimeError("lora_param must alias engine.lora")
        ctx.engine = engine
        return engine.forward_train(img)

    @staticmethod
    def backward(ctx, dkey):
        g = ctx.engine.backward(dkey.contiguous())
        return None, g.clone(), None
