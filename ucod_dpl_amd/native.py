"""ctypes binding of libucod_dpl.so (C ABI in include/ucod_dpl.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor is not on a
gfx950 device, the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_native", "libucod_dpl.so")
# Experiment builds (`make -C ucod_dpl_amd/csrc variant ...`) are selected by tools/ with UCOD_DPL_EXPERIMENT_LIB; the variable is honoured
# only together with UCOD_DPL_ALLOW_EXPERIMENT=1, so that a stray environment variable cannot swap the product library.
if os.environ.get("UCOD_DPL_LIB"):
    raise RuntimeError("UCOD_DPL_LIB is no longer read (a stray variable must not swap the product library): use "
                       "UCOD_DPL_ALLOW_EXPERIMENT=1 UCOD_DPL_EXPERIMENT_LIB=<path> for an experiment build")
if os.environ.get("UCOD_DPL_EXPERIMENT_LIB") and os.environ.get("UCOD_DPL_ALLOW_EXPERIMENT") == "1":
    LIB_PATH = os.environ["UCOD_DPL_EXPERIMENT_LIB"]
ABI_VERSION = 5                                            # include/ucod_dpl.h: UCOD_ABI_VERSION

EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_SCALE_RESID_F32, EPI_PATCH_TOKENS_F32, EPI_KEY_NCHW_F32, EPI_BIAS_F32 = range(6)
EPI_GELU_BWD_BF16, EPI_BIAS_GELU_SAVE_BF16 = 6, 7          # ucod_gemm_bf16_train only (backbone-backward mode)
EPI_QKV_FP8 = 8                                            # QKV projection of the fp8 attention path
EPI_BIAS_SCALE_RESID_H16, EPI_PATCH_TOKENS_H16 = 9, 10     # f16 residual stream (VitDesc.resid16)
EPI_LNFOLD_BIAS_BF16, EPI_LNFOLD_GELU_BF16 = 11, 12        # ucod_gemm_lnfold only (LayerNorm folded into QKV / fc1; the fp16-operand build)
EPI_BIAS_SCALE_RESID_H16_STATS, EPI_PATCH_TOKENS_H16_STATS = 13, 14   # ucod_gemm_bf16_stats only (the producers that leave row partials)
EPI_BIAS_GELU_SPLIT2 = 15                                  # fc1 + GELU + two-term split of the result in one launch (the split-operand pass; bf16 library)
VIT_LAYER_STRIDE = 16
VIT_TRAIN_STRIDE = 7
LORA_AUG = 64

vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class VitDesc(C.Structure):
    _fields_ = [(n, ci) for n in ("B", "C", "H", "W", "P", "D", "heads", "F", "L", "Kpad")] + [("eps", cf)] + \
               [(n, ci) for n in ("full_last_layer", "gemm_variant", "attn_variant", "resid16", "ln_fold")]


class VitTrainDesc(C.Structure):
    _fields_ = [("vit", VitDesc), ("lora_r", ci), ("lora_scaling", cf), ("lora_dropout", cf), ("seed", C.c_ulonglong)]


class LoraDropout(C.Structure):
    _fields_ = [("p", cf), ("seed", C.c_ulonglong), ("layer", ci)]


class DiscParams(C.Structure):
    _fields_ = [(n, vp) for n in ("w1", "g1", "b1", "w2", "g2", "b2", "w3", "g3", "b3", "lin_w", "lin_b",
                                  "rm1", "rv1", "rm2", "rv2", "rm3", "rv3", "nbt")]


class DiscGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("w1", "g1", "b1", "w2", "g2", "b2", "w3", "g3", "b3", "lin_w", "lin_b")]


# name -> (restype, argtypes); must list EVERY symbol include/ucod_dpl.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "ucod_abi_version": (ci, []),
    "ucod_device_is_gfx950": (ci, []),
    "ucod_half_name": (C.c_char_p, []),
    "ucod_prof_enable": (ci, [ci]),
    "ucod_prof_num_classes": (ci, []),
    "ucod_prof_class_name": (C.c_char_p, [ci]),
    "ucod_prof_collect": (ci, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "ucod_clock_probe": (ci, [vp, vp]),
    "ucod_split_products": (ci, [ci]),
    "ucod_split_rows": (ci, [vp, C.c_long, vp, ci, ci, ci, ci, ci, cf, vp]),
    "ucod_layernorm_split": (ci, [vp, vp, vp, vp, ci, ci, cf, ci, ci, vp]),
    "ucod_patch_im2col_split": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_attention_split_operand_bytes": (sz, [ci, ci, ci, ci]),
    "ucod_qkv_split": (ci, [vp, vp, ci, ci, ci, ci, cf, vp]),
    "ucod_attention_split_fwd": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "ucod_vit_split_workspace_bytes": (sz, [C.POINTER(VitDesc), ci]),
    "ucod_vit_split_stream_offset": (sz, [C.POINTER(VitDesc), ci]),
    "ucod_vit_forward_split": (ci, [C.POINTER(VitDesc), ci, C.POINTER(vp), vp, vp, vp, sz, vp]),
    "ucod_gemm_bf16": (ci, [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_gemm_lnfold": (ci, [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, cf, vp, ci, vp]),
    "ucod_gemm_bf16_stats": (ci, [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, vp, ci, vp]),
    "ucod_cls_rows_h16_stats": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ucod_row_stats_h16": (ci, [vp, vp, ci, ci, cf, vp]),
    "ucod_gemm_reload_tuning": (None, []),
    "ucod_resid16_overflow_fetch": (ci, [vp, vp]),
    "ucod_resid16_overflow_reset": (ci, [vp]),
    "ucod_resid16_overflow_bind": (ci, [vp]),
    "ucod_layernorm": (ci, [vp, vp, vp, vp, ci, ci, cf, ci, vp]),
    "ucod_layernorm_h16": (ci, [vp, vp, vp, vp, ci, ci, cf, vp]),
    "ucod_attention_fwd": (ci, [vp, vp, ci, ci, ci, cf, ci, vp]),
    "ucod_attention_fp8_workspace_bytes": (sz, [ci, ci, ci]),
    "ucod_attention_fwd_fp8": (ci, [vp, vp, vp, sz, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_attention_fp8_zero_pad": (ci, [vp, ci, ci, ci, vp]),
    "ucod_attention_fwd_fp8_fused": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_patch_im2col": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_cls_rows": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_cls_rows_h16": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_fill_qscale": (ci, [vp, ci, cf, vp]),
    "ucod_fill_qscale3": (ci, [vp, ci, cf, cf, cf, vp]),
    "ucod_cast_f32_bf16": (ci, [vp, vp, sz, vp]),
    "ucod_vit_workspace_bytes": (sz, [C.POINTER(VitDesc)]),
    "ucod_vit_forward": (ci, [C.POINTER(VitDesc), C.POINTER(vp), vp, vp, vp, sz, vp]),
    "ucod_vit_train_workspace_bytes": (sz, [C.POINTER(VitTrainDesc)]),
    "ucod_vit_forward_train": (ci, [C.POINTER(VitTrainDesc), C.POINTER(vp), C.POINTER(vp), vp, vp, vp, sz, vp]),
    "ucod_vit_backward": (ci, [C.POINTER(VitTrainDesc), C.POINTER(vp), C.POINTER(vp), vp, vp, sz, vp]),
    "ucod_vit_lora_infer_workspace_bytes": (sz, [C.POINTER(VitTrainDesc)]),
    "ucod_vit_forward_lora_infer": (ci, [C.POINTER(VitTrainDesc), C.POINTER(vp), C.POINTER(vp), vp, vp, vp, sz, vp]),
    "ucod_gemm_bf16_train": (ci, [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, ci, vp]),
    "ucod_layernorm_lora": (ci, [vp, vp, vp, vp, ci, vp, ci, ci, cf, C.POINTER(LoraDropout), vp]),
    "ucod_layernorm_lora_h16": (ci, [vp, vp, vp, vp, ci, vp, ci, ci, cf, C.POINTER(LoraDropout), vp]),
    "ucod_layernorm_bwd_lora": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, vp, vp, ci, C.POINTER(LoraDropout), vp]),
    "ucod_layernorm_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, vp]),
    "ucod_layernorm_bwd_ex": (ci, [vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, cf, vp]),
    "ucod_layernorm_bwd_lora_ex": (ci, [vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, cf, vp, vp, ci, C.POINTER(LoraDropout), vp]),
    "ucod_attention_fwd_lse": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_attention_bwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ucod_key_grad_tokens": (ci, [vp, vp, ci, ci, ci, vp]),
    "ucod_lora_pack": (ci, [vp, ci, cf, vp, vp, ci, ci, vp]),
    "ucod_lora_grad_workspace_bytes": (sz, [ci]),
    "ucod_lora_grad": (ci, [vp, vp, vp, ci, cf, vp, ci, vp, sz, ci, ci, C.POINTER(LoraDropout), vp]),
    "ucod_bilinear_resize": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "ucod_bilinear_resize_adjoint": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "ucod_dba_project": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ucod_dba_project_split_workspace_bytes": (sz, [ci, ci]),
    "ucod_dba_project_split": (ci, [vp, vp, vp, vp, vp, sz, ci, ci, ci, ci, vp]),
    "ucod_dba_colnorm": (ci, [vp, ci, ci, vp, vp, ci, ci, vp]),
    "ucod_dba_heads_fwd": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_orth_workspace_bytes": (sz, [ci, ci]),
    "ucod_orth_gram_fwd": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_dba_bwd_workspace_bytes": (sz, [ci, ci]),
    "ucod_dba_bwd": (ci, [vp, ci, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_dba_wgrad": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_dba_wgrad_split": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_disc_saved_bytes": (sz, [ci, ci]),
    "ucod_disc_fwd": (ci, [vp, C.POINTER(DiscParams), vp, vp, ci, ci, ci, vp]),
    "ucod_disc_bwd_workspace_bytes": (sz, [ci, ci]),
    "ucod_disc_bwd": (ci, [vp, C.POINTER(DiscParams), vp, vp, C.POINTER(DiscGrads), ci, vp, ci, ci, vp]),
    "ucod_unfold3x3": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_bn_lrelu_workspace_bytes": (sz, [ci]),
    "ucod_bn_lrelu_train": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, ci, vp, sz, vp]),
    "ucod_linear_sigmoid": (ci, [vp, vp, vp, vp, ci, ci, vp]),
    "ucod_bn_lrelu_train_save": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, cf, ci, vp, sz, vp]),
    "ucod_bn_lrelu_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, ci, vp, sz, vp]),
    "ucod_fold3x3": (ci, [vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "ucod_conv_wgrad_f32": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ucod_linear_sigmoid_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ucod_disc_bce": (ci, [vp, vp, vp, vp, vp, ci, cf, vp]),
    "ucod_apm_bce": (ci, [vp, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_binarize": (ci, [vp, vp, sz, ci, vp]),
    "ucod_ccl8_host": (ci, [vp, ci, ci, vp]),
    "ucod_pil_resize_u8_host": (ci, [vp, ci, ci, vp, ci, ci, ci]),
    "ucod_vit_last_ln1_offset": (sz, [C.POINTER(VitDesc)]),
    "ucod_cls_qk": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ucod_cls_attention": (ci, [vp, vp, vp, vp, ci, ci, ci, cf, vp]),
    "ucod_bkg_seg": (ci, [vp, vp, cf, cf, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ucod_ccl8_workspace_bytes": (sz, [ci, ci]),
    "ucod_ccl8_components": (ci, [vp, ci, ci, vp, ci, vp, vp, sz, vp]),
    "ucod_paste_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "ucod_paste_resized_u8": (ci, [vp, ci, ci, ci, vp, vp, ci, ci, vp, sz, vp]),
    "ucod_paste_resized_u8_multi": (ci, [vp, ci, ci, ci, vp, vp, vp, ci, ci, ci, vp, sz, vp]),
    "ucod_crop_resize_norm_multi": (ci, [vp, vp, ci, vp, vp, ci, vp, ci, ci, vp, sz, vp]),
    "ucod_crop_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "ucod_crop_resize_norm": (ci, [vp, ci, ci, vp, ci, vp, ci, ci, vp, sz, vp]),
    "ucod_cross_attention96_fwd": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, ci, ci, vp]),
    "ucod_gather_tokens": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ucod_entropy_scores": (ci, [vp, ci, vp, vp, ci, ci, ci, ci, vp]),
    "ucod_dwconv7_maskdec": (ci, [vp, vp, vp, vp, cf, vp, ci, ci, ci, ci, vp]),
    "ucod_window_scatter": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ucod_window_loss": (ci, [vp, vp, vp, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ucod_gated_ensemble_workspace_bytes": (sz, [ci, ci, ci]),
    "ucod_gated_ensemble": (ci, [vp, vp, vp, vp, vp, cf, vp, vp, vp, ci, ci, ci, vp]),
    "ucod_step_loss": (ci, [vp, vp, ci, vp, vp]),
    "ucod_copy_segments": (ci, [vp, vp, vp, ci, vp]),
    "ucod_zero_segments": (ci, [vp, vp, ci, vp]),
    "ucod_accumulators_prezeroed": (ci, [ci]),
    "ucod_adamw_ema": (ci, [vp, vp, vp, vp, vp, sz, cf, cf, cf, cf, cf, ci, cf, vp]),
    "ucod_cod_metrics_workspace_bytes": (sz, [ci, ci, ci]),
    "ucod_cod_metrics": (ci, [vp, vp, ci, ci, ci, vp, vp, sz, vp]),
}
COD_RECORD = 1032

LIB_PATH_F16 = os.path.join(_HERE, "_native", "libucod_dpl_f16.so")   # same sources built with -DUCOD_HALF_F16 (fp16 forward-path operands)
if os.environ.get("UCOD_DPL_EXPERIMENT_LIB_F16") and os.environ.get("UCOD_DPL_ALLOW_EXPERIMENT") == "1":
    LIB_PATH_F16 = os.environ["UCOD_DPL_EXPERIMENT_LIB_F16"]          # tools/: e.g. the -DUCOD_LAB_KNOBS build (`make -C ucod_dpl_amd/csrc knobs`)
_libs = {}


def load(half="bf16"):
    """Load a shared library (once each).  ``half`` selects the build: "bf16" (default, the product path of BASELINE configs[1]) or
    "f16" (the same kernels on IEEE fp16 operands -- the arithmetic type of the reference's fp16-autocast launcher).
    Raises if the library has not been built -- never falls back."""
    if half not in ("bf16", "f16"):
        raise ValueError(f"half must be 'bf16' or 'f16', got {half!r}")
    if half not in _libs:
        path = LIB_PATH if half == "bf16" else LIB_PATH_F16
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C ucod_dpl_amd/csrc` (hipcc --offload-arch=gfx950). There is no fallback path.")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.ucod_abi_version() != ABI_VERSION:
            raise ImportError(f"{path} has ABI version {lib.ucod_abi_version()}, this binding needs {ABI_VERSION}: rebuild it "
                              f"(`make -C ucod_dpl_amd/csrc`)")
        if lib.ucod_half_name().decode() != half:
            raise ImportError(f"{path} was built for {lib.ucod_half_name().decode()} operands, expected {half}")
        _libs[half] = lib
    return _libs[half]


# ---- the laboratory library (csrc/variants/*.hip, `make -C ucod_dpl_amd/csrc variants`): never loaded by the product path
LIB_PATH_LAB = os.path.join(_HERE, "_native", "libucod_dpl_variants.so")
LAB_SIGNATURES = {
    "ucod_attention_fwd_lab": (ci, [vp, vp, ci, ci, ci, cf, ci, vp]),
    "ucod_attention_fwd_asm_lab": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),     # the hand-placed assembly kernels (form 0 = pw64, 1 = pw32), optional LSE
    "ucod_gemm_bf16_lab": (ci, [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]),
    "ucod_gemm_bf16_asm_lab": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]),
    "ucod_gemm_bf16_asm_lab_forms": (ci, []),
    "ucod_gemm_bf16_asm_lab_label": (C.c_char_p, [ci]),
}
ATTN_PRODUCT_VARIANTS = (0, 2, 5, 66)           # 5 / 66 = attn_fwd_v5_kernel / attn_fwd_v6_kernel by name
GEMM_PRODUCT_VARIANTS = (0, 1, 2, 9, 10, 12, 13, 14)


def have_lab():
    return os.path.exists(LIB_PATH_LAB)


def load_lab():
    """The experiment variants of rounds 1-2 (tools/, tests marked `variants`).  Raises if `make variants` has not been run."""
    if "lab" not in _libs:
        if not have_lab():
            raise ImportError(f"{LIB_PATH_LAB} is missing: build it with `make -C ucod_dpl_amd/csrc variants` (laboratory kernels; "
                              f"the product path never needs it)")
        lib = C.CDLL(LIB_PATH_LAB)
        for name, (res, args) in LAB_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _libs["lab"] = lib
    return _libs["lab"]


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("ucod_dpl_amd: tensor is not on a GPU; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("ucod_dpl_amd: tensor must be contiguous")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (invalid argument)" if rc == -1 else ""))
