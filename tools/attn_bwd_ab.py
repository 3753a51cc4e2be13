#!/usr/bin/env python3
"""A/B of the attention backward (ucod_attention_bwd) at the C2 shape in ONE process: the product library against another build of the
same entry point (default: the round-2 kernels, ucod_dpl_amd/_native/libucod_attn_bwd_r2.so, built from `git show <r2>:.../attention_bwd.hip`).
Interleaved rounds, median / min per arm, outputs compared bit for bit."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import native as N  # noqa: E402

B, tok, heads = (int(x) for x in os.environ.get("ATTN_SHAPE", "32,1370,12").split(","))
ROUNDS, ITERS = int(os.environ.get("ATTN_ROUNDS", "5")), int(os.environ.get("ATTN_ITERS", "6"))
other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ucod_dpl_amd", "_native", "libucod_attn_bwd_r2.so")
lib = N.load()
lib2 = C.CDLL(other)
vp, ci = C.c_void_p, C.c_int
lib2.ucod_attention_bwd.restype, lib2.ucod_attention_bwd.argtypes = ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
D = heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv_f = torch.randn(B * tok, 3 * D, device="cuda", generator=g)
qkv_f[:, :D] *= 0.125 * 1.4426950408889634
qkv = qkv_f.to(torch.bfloat16)
out = torch.empty(B * tok, D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B * heads * tok, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
N.check(lib.ucod_attention_fwd_lse(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, tok, heads, st), "fwd_lse")
dout = (torch.randn(B * tok, D, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
ldd = 3 * D + 64
arms = [("product", lib.ucod_attention_bwd), ("other", lib2.ucod_attention_bwd)]
res, times = {}, {a[0]: [] for a in arms}
for name, fn in arms:
    dq = torch.zeros(B * tok, ldd, dtype=torch.bfloat16, device="cuda")
    delta = torch.zeros(B * heads * tok, dtype=torch.float32, device="cuda")
    assert fn(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), ldd, B, tok, heads, st) == 0
    torch.cuda.synchronize()
    res[name] = (dq, delta)
for r in range(ROUNDS):
    for name, fn in arms:
        dq, delta = res[name]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), ldd, B, tok, heads, st)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / ITERS * 1e3)
fl = 10.0 * B * heads * tok * tok * 64
print(f"# attention backward A/B at B={B}, N={tok}, heads={heads} (other = {os.path.basename(other)}): {ROUNDS} interleaved rounds x {ITERS}; us per call (dQ + dKV kernels)")
for name, _ in arms:
    t = times[name]
    print(f"{name:8s}: median {statistics.median(t):7.1f} us  min {min(t):7.1f} us   {fl / (statistics.median(t) * 1e-6) / 1e12:6.1f} TF/s algorithmic (5 products) "
          f"= {fl / (statistics.median(t) * 1e-6) / 2.5e15:.3f} of 2.5 PF")
print("dqkv bitwise equal:", bool(torch.equal(res["product"][0][:, :3 * D], res["other"][0][:, :3 * D])), " delta equal:", bool(torch.equal(res["product"][1], res["other"][1])))
