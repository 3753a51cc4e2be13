#!/usr/bin/env python3
"""Functional simulation of the generated GEMM kernel (gen_gemm.py) on a small problem: one workgroup (8 waves) walks all tiles with stride 1.
Checks the result against numpy, the LDS-DMA / barrier / waitcnt protocol (sim.py) and the static wait-state audit (checks.py)."""
import sys
import struct
import numpy as np
from .gen_gemm import GemmGen, KERNEL_NAME, KARG_BYTES
from .sim import Machine, bf16_round, bf16_to_f32
from .checks import audit


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    import os
    g = GemmGen(K=K, stride=1, sync=int(os.environ.get("GEMM_SYNC", "4")))
    prog = g.build()
    v = audit(prog)
    for x in v[:30]:
        print("AUDIT", x)
    print(len(v), "audit violations;", prog.count(), "instructions")
    rng = np.random.default_rng(0)
    Af = rng.standard_normal((M, K)).astype(np.float32)
    Wf = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Ab, Wb = bf16_round(Af).astype(np.uint16), bf16_round(Wf).astype(np.uint16)
    m = Machine(prog, nwaves=8, lds_bytes=g.lds_bytes)
    pA, pW, pB = m.alloc(Ab.nbytes), m.alloc(Wb.nbytes), m.alloc(bias.nbytes)
    pO = m.alloc(M * N * 2 + 4096)
    m.write(pA, Ab); m.write(pW, Wb); m.write(pB, bias)
    guard = np.full(M * N + 2048, 0x7FC1, dtype=np.uint16)
    m.write(pO, guard)
    tiles_m, tiles_n = (M + 255) // 256, N // 256
    ntiles = tiles_m * tiles_n
    magic = 0 if tiles_n == 1 else ((1 << 32) + tiles_n - 1) // tiles_n
    add1 = 1 if tiles_n == 1 else 0
    ka = m.alloc(KARG_BYTES)
    m.write(ka, np.frombuffer(struct.pack("<QQQQiiiiIiiiQ", pA, pW, pB, pO, M, N, tiles_n, ntiles, magic, add1, ntiles, 0, 0), dtype=np.uint8))

    def setup(w):
        w.s[0], w.s[1] = ka & 0xFFFFFFFF, ka >> 32
        w.s[2] = 0
        w.v[0] = np.arange(64, dtype=np.uint32) + 64 * w.wid
    steps = m.run(KERNEL_NAME, setup)
    out = m.read(pO, (M * N + 2048) * 2).view(np.uint16)
    got = bf16_to_f32(out[:M * N].astype(np.uint32)).reshape(M, N)
    ref = bf16_to_f32(Ab.astype(np.uint32)).astype(np.float64) @ bf16_to_f32(Wb.astype(np.uint32)).astype(np.float64).T + bias
    err = np.abs(got - ref)
    print(f"M {M} N {N} K {K}: steps {steps} mfma {m.mfma_count}  max|err| {err.max():.4g} (|ref| max {np.abs(ref).max():.3g})  untouched-guard {np.all(out[M * N:] == 0x7FC1)}")
    bad = np.argwhere(err > 0.02 * np.abs(ref).max())
    print("bad outputs:", len(bad), bad[:8].tolist())
    for x in m.violations[:30]:
        print("VIOL", x)
    print(len(m.violations), "protocol violations")


if __name__ == "__main__":
    main()
