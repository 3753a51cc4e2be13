#!/usr/bin/env python3
"""Timing of generator variants of the assembly attention kernel in ONE process, interleaved rounds (GPU box only): each variant is generated,
assembled with the ROCm clang, loaded through the HIP module API and launched on torch's stream.  `full` is checked against
attn_fwd_v5_kernel; the ablation variants (a component removed: wrong results, timing only) say where a tile's cycles go.

usage: python tools/attn_asm/bench_variants.py [name=flag+flag ...]     e.g.  full= nosm=nosm+nobranch mfma_only=nosm+nolds+nodma+nobar+nodetect
"""
import ctypes as C
import math
import os
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tools.attn_asm.gen_attn import kernel_text, KARG_BYTES  # noqa: E402
from tools.attn_asm import gen_attn32  # noqa: E402
from tools.attn_asm.run_sim import kernargs  # noqa: E402
from ucod_dpl_amd import ops  # noqa: E402

LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin")
B, tok, heads = (int(x) for x in os.environ.get("ATTN_SHAPE", "32,1370,12").split(","))
ROUNDS, ITERS = int(os.environ.get("ATTN_ROUNDS", "5")), int(os.environ.get("ATTN_ITERS", "10"))
hip = C.CDLL("libamdhip64.so")
tmp = tempfile.mkdtemp(prefix="attnasm_")


def build(name, pw32=False, **kw):
    txt, g = (gen_attn32.kernel_text if pw32 else kernel_text)("bf16", name="k_" + name, **kw)
    s, o, co = (os.path.join(tmp, name + e) for e in (".s", ".o", ".co"))
    open(s, "w").write(txt)
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", o, "-o", co])
    mod, fn = C.c_void_p(), C.c_void_p()
    assert hip.hipModuleLoad(C.byref(mod), co.encode()) == 0
    assert hip.hipModuleGetFunction(C.byref(fn), mod, ("k_" + name).encode()) == 0
    return fn


def launcher(fn, qkv, out, lse_ptr=0, dbg_ptr=0, threads=256):
    npairs = B * heads
    nqb = (tok + 255) // 256
    stride = min(32, ((npairs + 7) // 8) * nqb)
    ka = kernargs(qkv.data_ptr(), out.data_ptr(), lse_ptr, tok, heads, npairs, stride, dbg_ptr)
    buf = (C.c_uint32 * len(ka))(*[int(x) for x in ka])
    size = C.c_size_t(KARG_BYTES)
    cfg = (C.c_void_p * 5)(1, C.cast(buf, C.c_void_p), 2, C.cast(C.pointer(size), C.c_void_p), 3)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def go():
        rc = hip.hipModuleLaunchKernel(fn, 8 * stride, 1, 1, threads, 1, 1, 0, st, None, cfg)
        assert rc == 0, rc
    go.keep = (buf, size, cfg)
    return go


def main():
    specs = sys.argv[1:] or ["full="]
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv_f = torch.randn(B * tok, 3 * heads * 64, device="cuda", generator=g)
    qkv_f[:, :heads * 64] *= 0.125 * math.log2(math.e)
    qkv = qkv_f.to(torch.bfloat16)
    if os.environ.get("ATTN_ZERO"):
        qkv.zero_()                       # constant operands: the clock the chip holds when the data does not toggle
    ref = ops.attention(qkv, B, tok, heads, scale=0.0, variant=5)
    arms, dbgs = {}, {}
    for sp in specs:
        name, flags = sp.split("=", 1)
        kw = {}
        fl = [f for f in flags.split("+") if f]
        pw32 = "pw32" in fl
        abl = [f for f in fl if not f.startswith(("thr", "dmagap", "ud", "margin", "ring", "pw32"))]
        for f in fl:
            if f.startswith("thr"):
                kw["thr_exp"] = int(f[3:])
            if f.startswith("dmagap"):
                kw["dma_gap"] = int(f[6:])
            if f.startswith("ud"):
                kw["unit_detect"] = bool(int(f[2:]))
            if f.startswith("ring"):
                kw["ring"] = int(f[4:])
            if f.startswith("margin"):
                kw["margin"] = int(f[6:])
        out = torch.zeros(B * tok, heads * 64, dtype=torch.bfloat16, device="cuda")
        dbg = torch.zeros(256 * 4 * 8, dtype=torch.int32, device="cuda") if "stamps" in abl else None
        arms[name] = (launcher(build(name, pw32=pw32, abl=abl, **kw), qkv, out, dbg_ptr=dbg.data_ptr() if dbg is not None else 0,
                               threads=512 if pw32 else 256), out, abl)
        if dbg is not None:
            dbgs[name] = (dbg, pw32)
    arms["v5"] = (lambda: ops.attention(qkv, B, tok, heads, scale=0.0, variant=5), ref, ["ref"])
    for name, (go, out, abl) in arms.items():
        go()
    torch.cuda.synchronize()
    times = {n: [] for n in arms}
    for r in range(ROUNDS):
        for name, (go, out, abl) in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                go()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / ITERS * 1e3)
    fl = 4.0 * B * heads * tok * tok * 64
    print(f"# B={B} N={tok} heads={heads}: {ROUNDS} interleaved rounds x {ITERS} launches, us per launch")
    for name, (go, out, abl) in arms.items():
        med, mn = statistics.median(times[name]), min(times[name])
        d = (out.float() - ref.float()).abs().max().item()
        print(f"{name:22s} median {med:7.1f}  min {mn:7.1f}  {fl / (med * 1e-6) / 2.5e15:.3f} of 2.5 PF   max|out - v5| {d:.3g}   [{'+'.join(abl)}]", flush=True)


    for name, (dbg, pw32) in dbgs.items():
        arms[name][0]()
        torch.cuda.synchronize()
        if pw32:
            d = dbg.cpu().view(-1, 4).double()
            d = d[d[:, 3] > 0]
            per = d[:, :3] / d[:, 3:4]
            print(f"stamps {name}: cycles per steady iteration of a live wave, mean over {len(d)} waves: step a {per[:, 0].mean().item():.0f}  step b {per[:, 1].mean().item():.0f}"
                  f"  wait+barrier {per[:, 2].mean().item():.0f}  total {per.sum(1).mean().item():.0f}")
            continue
        d = dbg.cpu().view(-1, 8).double()
        d = d[d[:, 5] > 0]
        per = d[:, :5] / d[:, 5:6]
        print(f"stamps {name}: cycles per steady iteration, mean over {len(d)} waves: steps a b c d = " + " ".join(f"{x:.0f}" for x in per[:, :4].mean(0).tolist())
              + f"  wait+barrier {per[:, 4].mean().item():.0f}  total {per.sum(1).mean().item():.0f}  (min wave {per.sum(1).min().item():.0f}, max {per.sum(1).max().item():.0f})")
        for w in range(4):
            pw = per[w::4]
            print(f"   wave {w}: " + " ".join(f"{x:.0f}" for x in pw.mean(0).tolist()))


if __name__ == "__main__":
    main()
