#!/usr/bin/env python3
"""Run the generated attention kernel in the functional simulator against a float64 softmax reference (CPU only).
usage: python -m tools.attn_asm.run_sim [B heads N [stride [dtype]]]"""
import math
import sys
import numpy as np
from .gen_attn import Gen, KERNEL_NAME, KARG_BYTES
from . import gen_attn32
from .sim import Machine, bf16_round, bf16_to_f32, f16_round, f16_to_f32, U32, F32


def magic(d):
    return (0, 1) if d == 1 else (((1 << 32) + d - 1) // d, 0)


def kernargs(qkv, out, lse, N, heads, npairs, stride, dbg=0):
    nqb, nt = (N + 255) // 256, (N + 63) // 64
    mq, aq = magic(nqb)
    mh, ah = magic(heads)
    ka = np.zeros(KARG_BYTES // 4, dtype=np.uint32)
    for k, ptr in enumerate((qkv, out, lse)):
        ka[2 * k], ka[2 * k + 1] = ptr & 0xFFFFFFFF, ptr >> 32
    ka[6:16] = [N, heads, npairs, nqb, mq, mh, nt, stride, aq, ah]
    ka[16], ka[17] = dbg & 0xFFFFFFFF, dbg >> 32
    return ka


def reference(qkv_f, B, N, heads):
    D = heads * 64
    x = qkv_f.reshape(B, N, 3, heads, 64).astype(np.float64)
    q, k, v = x[:, :, 0].transpose(0, 2, 1, 3), x[:, :, 1].transpose(0, 2, 1, 3), x[:, :, 2].transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2)
    m = s.max(-1, keepdims=True)
    p = np.exp2(s - m)
    l = p.sum(-1, keepdims=True)
    o = (p / l) @ v
    lse = (m + np.log2(l))[..., 0]
    return o.transpose(0, 2, 1, 3).reshape(B * N, D), lse          # lse [B, heads, N]


def simulate(B, heads, N, stride=32, dtype="bf16", seed=0, qkv_f=None, thr_exp=None, with_lse=True, verbose=False, kernel="pw64", **genkw):
    if kernel == "pw64":
        g, nwaves, entry = Gen(dtype=dtype, thr_exp=thr_exp, **genkw), 4, KERNEL_NAME
    else:
        g, nwaves, entry = gen_attn32.Gen32(dtype=dtype, **genkw), 8, gen_attn32.KERNEL_NAME
    prog = g.build()
    D = heads * 64
    rng = np.random.default_rng(seed)
    if qkv_f is None:
        qkv_f = rng.standard_normal((B * N, 3 * D)).astype(F32) * 1.5
        qkv_f[:, :D] *= 0.125 * math.log2(math.e)
    rnd, back = (bf16_round, bf16_to_f32) if dtype == "bf16" else (f16_round, f16_to_f32)
    bits = rnd(qkv_f.astype(F32).ravel()).astype(np.uint16)
    qkv_q = back(bits.astype(U32)).reshape(B * N, 3 * D)
    m = Machine(prog, dtype=dtype, nwaves=nwaves)
    a_qkv = m.alloc(bits.nbytes + 4096)
    m.write(a_qkv, bits)
    a_out = m.alloc(B * N * D * 2 + 4096)
    m.mem[a_out:a_out + B * N * D * 2] = 0xAB
    a_lse = m.alloc(B * heads * N * 4 + 4096) if with_lse else 0
    a_ka = m.alloc(KARG_BYTES)
    npairs = B * heads
    m.write(a_ka, kernargs(a_qkv, a_out, a_lse, N, heads, npairs, stride))
    total = {"steps": 0, "mfma": 0}
    for wg in range(8 * stride):
        m.waves = [type(m.waves[0])(w) for w in range(nwaves)]
        m.lds[:] = 0xEE
        m.lds_inflight[:] = 0
        m.lds_pub_epoch[:] = -1
        m.lds_read_epoch[:] = -1
        m.epoch = 0

        def setup(w, wg=wg):
            w.s[0], w.s[1] = a_ka & 0xFFFFFFFF, a_ka >> 32
            w.s[2] = wg
            w.v[0] = np.arange(64, dtype=U32) + 64 * w.wid
        total["steps"] += m.run(entry, setup)
    total["mfma"] = m.mfma_count
    out_bits = m.read(a_out, B * N * D * 2).view(np.uint16)
    out = back(out_bits.astype(U32)).reshape(B * N, D)
    ref, lse_ref = reference(qkv_q, B, N, heads)
    res = {"max_abs": float(np.abs(out - ref).max()), "rel_l2": float(np.linalg.norm(out - ref) / np.linalg.norm(ref)),
           "violations": m.violations, "unwritten": int((out_bits == 0xABAB).sum()), **total}
    if with_lse:
        lse = m.read(a_lse, B * heads * N * 4).view(F32).reshape(B, heads, N)
        res["lse_max_abs"] = float(np.abs(lse - lse_ref).max())
    return res


if __name__ == "__main__":
    a = sys.argv[1:]
    B, heads, N = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1, 1, 200)
    stride = int(a[3]) if len(a) > 3 else 1
    dtype = a[4] if len(a) > 4 else "bf16"
    r = simulate(B, heads, N, stride=stride, dtype=dtype, kernel=a[5] if len(a) > 5 else "pw64")
    v = r.pop("violations")
    print(r)
    for x in v[:20]:
        print("VIOLATION", x)
