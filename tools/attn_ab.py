#!/usr/bin/env python3
"""A/B of attention-forward kernels at the C2 shape (B=32, 12 heads, N=1370) in ONE process, interleaved rounds (median and min per arm).

Arms are `label=library:variant` with library one of
    product   ucod_dpl_amd/_native/libucod_dpl.so           (ucod_attention_fwd)
    lab       ucod_dpl_amd/_native/libucod_dpl_variants.so  (ucod_attention_fwd_lab; `make -C ucod_dpl_amd/csrc variants`)
    <path>    an experiment build of the product library     (`make -C ucod_dpl_amd/csrc variant FILE=attention NAME=forms DEFS=-DUCOD_ATTN_LAB_FORMS`)
e.g.  python tools/attn_ab.py r2=lab:102 new=product:2 f1=ucod_dpl_amd/_native/libucod_dpl_forms.so:21
Every arm's output is compared with the first arm's (max abs difference) so that a fast wrong kernel shows."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import native as N  # noqa: E402

B, tok, heads = (int(x) for x in os.environ.get("ATTN_SHAPE", "32,1370,12").split(","))
ROUNDS, ITERS = int(os.environ.get("ATTN_ROUNDS", "7")), int(os.environ.get("ATTN_ITERS", "20"))
sig = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]


def arm(spec):
    label, rest = spec.split("=", 1)
    libname, variant = rest.rsplit(":", 1)
    if libname == "product":
        fn = N.load().ucod_attention_fwd
    elif libname == "lab":
        fn = N.load_lab().ucod_attention_fwd_lab
    else:
        lib = C.CDLL(os.path.join(ROOT, libname) if not os.path.isabs(libname) else libname)
        fn = lib.ucod_attention_fwd
        fn.restype, fn.argtypes = C.c_int, sig
    return label, fn, int(variant)


arms = [arm(s) for s in sys.argv[1:]] or [arm("product=product:2")]
g = torch.Generator(device="cuda").manual_seed(0)
qkv_f = torch.randn(B * tok, 3 * heads * 64, device="cuda", generator=g)
qkv_f[:, :heads * 64] *= 0.125 * 1.4426950408889634
qkv = qkv_f.to(torch.bfloat16)              # Q pre-scaled, as the QKV epilogue hands it over
outs = {}
fl = 4.0 * B * heads * tok * tok * 64
times = {a[0]: [] for a in arms}
st = torch.cuda.current_stream().cuda_stream
for label, fn, v in arms:
    o = torch.zeros(B * tok, heads * 64, dtype=torch.bfloat16, device="cuda")
    rc = fn(qkv.data_ptr(), o.data_ptr(), B, tok, heads, 0.0, v, st)
    assert rc == 0, (label, rc)
    torch.cuda.synchronize()
    outs[label] = o
for r in range(ROUNDS):
    for label, fn, v in arms:
        o = outs[label]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn(qkv.data_ptr(), o.data_ptr(), B, tok, heads, 0.0, v, st)
        e1.record()
        torch.cuda.synchronize()
        times[label].append(e0.elapsed_time(e1) / ITERS * 1e3)
first = arms[0][0]
print(f"# attention forward A/B at B={B}, N={tok}, heads={heads}: {ROUNDS} interleaved rounds x {ITERS} launches; us per launch")
for label, _, v in arms:
    t = times[label]
    med, mn = statistics.median(t), min(t)
    d = (outs[label].float() - outs[first].float()).abs().max().item()
    print(f"{label:12s} variant {v:3d}: median {med:7.1f} us  min {mn:7.1f} us  {fl / (med * 1e-6) / 1e12:7.1f} TF/s ({fl / (med * 1e-6) / 2.5e15:.3f} of 2.5 PF)   max|out - {first}| = {d:.3g}")
