#!/usr/bin/env python3
"""A/B of ucod_layernorm_bwd (+ the LoRA / dropout form) between the product library and another build (argv[1]) at the backbone-backward
shapes: rows = 16 x 1370 (one image-parallel chunk) and 32 x 1370, D = 768; interleaved rounds; outputs compared bitwise."""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import native as N  # noqa: E402

arms = [("product", N.load())]
for spec in sys.argv[1:]:
    label, path = spec.split("=", 1)
    lib = C.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
    for name in ("ucod_layernorm_bwd", "ucod_layernorm_bwd_lora"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = N.SIGNATURES[name]
    arms.append((label, lib))
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
D, r = 768, 2
for rows in (16 * 1370, 32 * 1370):
    dy = torch.randn(rows, D, device="cuda", generator=g)
    x = torch.randn(rows, D, device="cuda", generator=g) * 2 + 0.3
    gamma = torch.rand(D, device="cuda", generator=g) + 0.5
    dres = torch.randn(rows, D, device="cuda", generator=g)
    scale = torch.rand(D, device="cuda", generator=g) + 0.5
    dqkv = (torch.randn(rows, 3 * D + 64, device="cuda", generator=g)).to(torch.bfloat16)
    lora = torch.randn(6 * r * D, device="cuda", generator=g) * 0.1
    drop = N.LoraDropout()
    drop.p, drop.seed, drop.layer = 0.05, 1234567, 3
    moved = rows * D * (4 + 4 + 4 + 4 + 2)
    for form in ("plain", "lora"):
        outs, times = {}, {a[0]: [] for a in arms}

        def run(lib, dx, so):
            if form == "plain":
                return lib.ucod_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), dres.data_ptr(), scale.data_ptr(), dx.data_ptr(), so.data_ptr(), rows, D, 1e-6, st)
            return lib.ucod_layernorm_bwd_lora(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), dres.data_ptr(), scale.data_ptr(), dx.data_ptr(), so.data_ptr(), rows, D, 1e-6,
                                               dqkv.data_ptr(), lora.data_ptr(), r, C.byref(drop), st)
        for label, lib in arms:
            dx, so = torch.empty(rows, D, device="cuda"), torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
            assert run(lib, dx, so) == 0
            torch.cuda.synchronize()
            outs[label] = (dx, so)
        for rnd in range(7):
            for label, lib in arms:
                dx, so = outs[label]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(lib, dx, so)
                e1.record()
                torch.cuda.synchronize()
                times[label].append(e0.elapsed_time(e1) / 20 * 1e3)
        for label, _ in arms:
            med = statistics.median(times[label])
            eq = all(torch.equal(a, b) for a, b in zip(outs[label], outs[arms[0][0]]))
            print(f"rows {rows:6d} {form:5s} {label:8s}: {med:7.1f} us  {moved / (med * 1e-6) / 1e12:5.2f} TB/s moved   bitwise equal to product: {eq}")
