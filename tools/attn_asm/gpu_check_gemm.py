#!/usr/bin/env python3
"""First contact of the assembly GEMM (gen_gemm.py) with a GPU: small and ragged shapes against an f64 product and against the product kernel
(bitwise where the K order is the same), then interleaved timing against the product's large-tile kernels.  Run under `timeout`."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ucod_dpl_amd import ops  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "check"
DEV = "cuda"


def data(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    return x, w, b


if stage == "check":
    for (M, N) in [(256, 256), (300, 512), (1000, 768), (2048 + 64, 2304), (8 * 1370, 2304), (43840, 2304), (43840, 3072), (43840, 768)]:
        x, w, b = data(M, N, 768, M + N)
        ref = x.double() @ w.double().t() + b.double()
        pad = torch.full((M + 64, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        out = ops.linear_bf16_asm(x, w, b, out=pad[:M])
        torch.cuda.synchronize()
        prod = ops.linear_bf16(x, w, b)
        e = (out.double() - ref).abs().max().item()
        ep = (prod.double() - ref).abs().max().item()
        again = ops.linear_bf16_asm(x, w, b)
        neq = (out != prod).float().mean().item()
        print(f"M {M} N {N}: asm max|err| {e:.4g} (product {ep:.4g}; |ref| max {ref.abs().max().item():.3g})  differs-from-product {neq:.2e}  "
              f"repeat-bitwise {torch.equal(out, again)}  nan {torch.isnan(out.float()).any().item()}  rows-beyond-M-untouched {bool(torch.isnan(pad[M:].float()).all().item())}", flush=True)
elif stage == "time":
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for (M, N) in [(43520, 2304), (43840, 2304), (43840, 3072), (43840, 768)]:
        x, w, b = data(M, N, 768, 1)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)

        def t_of(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        rows = []
        for rnd in range(3):
            ta = t_of(lambda: ops.linear_bf16_asm(x, w, b, out=out))
            tp = t_of(lambda: ops.linear_bf16(x, w, b))
            rows.append((ta, tp))
        gf = 2.0 * M * N * 768 / 1e9
        print(f"M {M} N {N} K 768 ({gf:.1f} GF): " + "  ".join(f"asm {a:.1f} us / product {p_:.1f} us" for a, p_ in rows)
              + f"   best asm {min(r[0] for r in rows):.1f} ({gf / min(r[0] for r in rows) / 1e3 * 1e3:.0f} TF/s)  best product {min(r[1] for r in rows):.1f}", flush=True)
elif stage == "forms":
    from ucod_dpl_amd import native as N
    lab = N.load_lab()
    nf = lab.ucod_gemm_bf16_asm_lab_forms()
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    which = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] != "all" else [f for f in range(nf) if f != 9]     # (9: the stamps build writes through dbg)
    shapes = [(43520, 2304)] if len(sys.argv) <= 4 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[4].split(",")]
    for (M, Nn) in shapes:
        x, w, b = data(M, Nn, 768, 1)
        out = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
        ref = ops.linear_bf16(x, w, b)

        def t_of(fn):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        best = {f: 1e9 for f in which}
        bestp = 1e9
        for rnd in range(4):
            for f in which:
                best[f] = min(best[f], t_of(lambda: ops.linear_bf16_asm(x, w, b, out=out, form=f)))
            bestp = min(bestp, t_of(lambda: ops.linear_bf16(x, w, b)))
        print(f"M {M} N {Nn}: product {bestp:.1f} us", flush=True)
        for f in which:
            out.zero_()
            ops.linear_bf16_asm(x, w, b, out=out, form=f)
            same = torch.equal(out, ref)
            print(f"  form {f:2d} {best[f]:7.1f} us  equal-to-product {same}  {lab.ucod_gemm_bf16_asm_lab_label(f).decode()}", flush=True)
elif stage == "stamps":
    form = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    M, Nn = (43520, 2304) if len(sys.argv) <= 3 else tuple(int(v) for v in sys.argv[3].split("x"))
    x, w, b = data(M, Nn, 768, 1)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
    dbg = torch.zeros(256 * 2 * 8, dtype=torch.int32, device=DEV)
    for _ in range(3):
        dbg.zero_()
        ops.linear_bf16_asm(x, w, b, out=out, dbg=dbg, form=form)
    torch.cuda.synchronize()
    d = dbg.cpu().view(256, 2, 8).double()
    names = ["R0 work", "barrier after R", "M0 work", "barrier after M", "R1 work", "DMA wait", "M1 work", "seam"]
    tiles = -(-M // 256) * (Nn // 256) / 256.0
    for g in range(2):
        tot = d[:, g].sum(1)
        print(f"group {g}: cycles per workgroup mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); per tile ({tiles:.2f} tiles per workgroup):")
        for k, nme in enumerate(names):
            print(f"    {nme:18s} {d[:, g, k].mean() / tiles:9.0f}   ({100 * d[:, g, k].mean() / tot.mean():5.1f} %)  max over workgroups {d[:, g, k].max() / tiles:9.0f}")
elif stage == "one":                      # for rocprofv3: `one <form> [reps] [MxN]` launches one form (form -1: the product kernel)
    form = int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    M, Nn = (43520, 2304) if len(sys.argv) <= 4 else tuple(int(v) for v in sys.argv[4].split("x"))
    x, w, b = data(M, Nn, 768, 1)
    out = torch.empty(M, Nn, dtype=torch.bfloat16, device=DEV)
    for _ in range(reps):
        if form < 0:
            ops.linear_bf16(x, w, b)
        else:
            ops.linear_bf16_asm(x, w, b, out=out, form=form)
    torch.cuda.synchronize()
    print("done")
