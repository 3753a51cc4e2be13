#!/usr/bin/env python3
"""A/B of builds of the product GEMM (ucod_gemm_bf16) on the four backbone shapes of BASELINE configs[1] (M = 32 x 1370 rows) in ONE
process: interleaved rounds, median / min per arm and shape.  Arms: `label=path-to-libucod_dpl*.so` (default arm `product` = the
in-tree product library); experiment builds come from `make -C ucod_dpl_amd/csrc variant NAME=.. DEFS=-D..`.
  python tools/gemm_ab.py noearly=ucod_dpl_amd/_native/libucod_dpl_noearly.so
Every arm's output is compared with the first arm's."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import native as N  # noqa: E402

M = int(os.environ.get("GEMM_M", str(32 * 1370)))
ROUNDS, ITERS = int(os.environ.get("GEMM_ROUNDS", "7")), int(os.environ.get("GEMM_ITERS", "20"))
vp, ci = C.c_void_p, C.c_int
sig = [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]
arms = [("product", N.load().ucod_gemm_bf16)]
for spec in sys.argv[1:]:
    label, path = spec.split("=", 1)
    lib = C.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
    lib.ucod_gemm_bf16.restype, lib.ucod_gemm_bf16.argtypes = ci, sig
    arms.append((label, lib.ucod_gemm_bf16))
g = torch.Generator(device="cuda").manual_seed(0)
D, F = 768, 3072
# (name, epilogue, N, K, out dtype, needs resid)
shapes = [("qkv_bias_bf16", N.EPI_BIAS_BF16, 3 * D, D), ("fc1_gelu_bf16", N.EPI_BIAS_GELU_BF16, F, D),
          ("proj_resid_h16", N.EPI_BIAS_SCALE_RESID_H16, D, D), ("fc2_resid_h16", N.EPI_BIAS_SCALE_RESID_H16, D, F),
          ("proj_resid_f32", N.EPI_BIAS_SCALE_RESID_F32, D, D), ("fc2_resid_f32", N.EPI_BIAS_SCALE_RESID_F32, D, F)]
st = torch.cuda.current_stream().cuda_stream
print(f"# product GEMM A/B at M = {M}: {ROUNDS} interleaved rounds x {ITERS} launches; us per launch (TF/s, fraction of 2.5 PF)")
for name, epi, Nn, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(Nn, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Nn, device="cuda", generator=g)
    scale = torch.rand(Nn, device="cuda", generator=g) + 0.5
    odt = torch.float16 if epi == N.EPI_BIAS_SCALE_RESID_H16 else (torch.float32 if epi == N.EPI_BIAS_SCALE_RESID_F32 else torch.bfloat16)
    resid0 = torch.randn(M, Nn, device="cuda", generator=g).to(odt)
    outs, times = {}, {a[0]: [] for a in arms}

    def run(fn, out):
        rp = out.data_ptr() if epi in (N.EPI_BIAS_SCALE_RESID_H16, N.EPI_BIAS_SCALE_RESID_F32) else None
        sp = scale.data_ptr() if rp else None
        rc = fn(epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), M, Nn, K, bias.data_ptr(), sp, rp, None, 1370, 0, st)
        assert rc == 0, (name, rc)

    for label, fn in arms:
        o = resid0.clone()
        run(fn, o)
        torch.cuda.synchronize()
        outs[label] = o.clone()
    for r in range(ROUNDS):
        for label, fn in arms:
            o = outs[label]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                run(fn, o)                      # (in-place residual epilogues keep accumulating: timing only after the first call)
            e1.record()
            torch.cuda.synchronize()
            times[label].append(e0.elapsed_time(e1) / ITERS * 1e3)
    fl = 2.0 * M * Nn * K
    ref = None
    for label, fn in arms:
        o = resid0.clone()
        run(fn, o)
        torch.cuda.synchronize()
        if ref is None:
            ref = o
        med, mn = statistics.median(times[label]), min(times[label])
        print(f"{name:16s} {label:10s}: median {med:7.1f} us  min {mn:7.1f} us  {fl / (med * 1e-6) / 1e12:7.1f} TF/s ({fl / (med * 1e-6) / 2.5e15:.3f})  "
              f"bitwise equal to {arms[0][0]}: {bool(torch.equal(o, ref))}")
