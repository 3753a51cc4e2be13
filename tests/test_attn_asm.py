"""The hand-placed attention kernels (tools/attn_asm): CPU-side checks of the generated assembly.
  * the functional simulator runs each generated kernel on small shapes against a float64 softmax -- layouts, the software pipeline, item
    seams, the masked last tile, dead waves -- and checks the asynchronous-memory protocol (counted waits, LDS-DMA ring, barriers);
  * the static wait-state audit (hazards hipcc would have padded) finds nothing;
  * the committed .s files are what the generators produce (they are build inputs: ucod_dpl_amd/csrc/Makefile).
Reference for the arithmetic: transformers modeling_dinov2.py:153-179 / models/backbones/dino.py:96-120 (base-2 softmax of pre-scaled Q)."""
import math
import os
import numpy as np
import pytest

from tools.attn_asm import run_sim, gen_attn, gen_attn32
from tools.attn_asm.checks import audit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "ucod_dpl_amd", "csrc", "variants", "asm")


@pytest.mark.parametrize("kernel,dtype,tol", [("pw64", "bf16", 4e-3), ("pw64", "f16", 6e-4), ("pw32", "bf16", 4e-3)])
@pytest.mark.parametrize("B,heads,N,stride", [(1, 1, 129, 1), (1, 2, 300, 1), (1, 1, 200, 32)])
def test_simulated_kernel_matches_softmax(kernel, dtype, tol, B, heads, N, stride):
    """129 = three tiles (one steady iteration) with a 1-key last tile; 300 with stride 1 = two items per workgroup (the seam, the epilogue
    under the next item's first tile, Q prefetch); 200 = rows past N in the only item (dead waves of the 32-row form)."""
    r = run_sim.simulate(B, heads, N, stride=stride, dtype=dtype, kernel=kernel)
    assert not r["violations"], r["violations"][:3]
    assert r["rel_l2"] < tol and r["max_abs"] < 12 * tol, r
    assert r["lse_max_abs"] < (6e-3 if (kernel, dtype) == ("pw64", "f16") else 2e-4), r          # fp16 form: the denominator sums ROUNDED probabilities
    assert r["unwritten"] <= 4          # (a bf16 pattern equal to the poison value can occur by chance)


@pytest.mark.parametrize("kernel", ["pw64", "pw32"])
def test_simulated_score_spike_and_dominant_first_tile(kernel):
    """a late key 138 (log2 units) above everything before it, and a first tile that dominates: the margin form must stay finite and exact"""
    D, tok = 64, 400
    rng = np.random.default_rng(21)
    base = rng.standard_normal((tok, 3 * D)).astype(np.float32) * 0.3
    a = base.copy(); a[5, :D] = 2.0; a[333, D:2 * D] = 6.0
    c = base.copy(); c[:64, D:2 * D] += 3.0; c[:, :D] = 1.0
    for x in (a, c):
        x = x.copy()
        x[:, :D] *= 0.125 * math.log2(math.e)
        r = run_sim.simulate(1, 1, tok, stride=1, qkv_f=x, kernel=kernel)
        assert not r["violations"] and r["max_abs"] < 2e-3 and r["rel_l2"] < 4e-3, r


def test_simulated_unit_detection_rescales():
    """the fp16 form's per-unit detection: every threshold (always / sometimes / never rescale) gives the same answer"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((300, 192)).astype(np.float32)
    x[:, :64] *= 0.125 * math.log2(math.e) * 3.0
    res = [run_sim.simulate(1, 1, 300, stride=1, qkv_f=x, dtype="f16", thr_exp=t) for t in (-20, 2, 13)]
    assert res[0]["steps"] > res[2]["steps"]                    # the out-of-line path really ran
    for r in res:
        assert not r["violations"] and r["rel_l2"] < 8e-4, r


@pytest.mark.parametrize("gen", [lambda: gen_attn.Gen("bf16"), lambda: gen_attn.Gen("f16"), lambda: gen_attn32.Gen32("bf16")])
def test_wait_state_audit_is_clean(gen):
    assert audit(gen().build()) == []


def test_simulator_flags_a_missing_wait():
    """the protocol checks are live: drop the fragment waits and the simulator must object"""
    orig = gen_attn.Gen.wait_frag
    gen_attn.Gen.wait_frag = lambda self, buf: None
    try:
        r = run_sim.simulate(1, 1, 200, stride=1)
    finally:
        gen_attn.Gen.wait_frag = orig
    assert any("outstanding" in v for v in r["violations"])


def _gemm_text():
    from tools.attn_asm import gen_gemm
    return gen_gemm.kernel_text(768, name="ucod_gemm_pk_k768_f0")[0]


@pytest.mark.parametrize("fname,text", [("attn_fwd_pw64_bf16.s", lambda: gen_attn.kernel_text("bf16")[0]),
                                        ("attn_fwd_pw32_bf16.s", lambda: gen_attn32.kernel_text("bf16")[0]),
                                        ("gemm_pk_k768_bf16.s", _gemm_text)])
def test_committed_assembly_is_current(fname, text):
    with open(os.path.join(ASM, fname)) as f:
        committed = f.read().split("\n", 1)[1]          # first line: the GENERATED banner
    assert committed == text(), f"{fname} is stale: run `make -C ucod_dpl_amd/csrc variants`"


@pytest.mark.parametrize("sync", [4, 1])
def test_gemm_generator_simulates_correctly(sync):
    """the hand-placed persistent GEMM (tools/attn_asm/gen_gemm.py, laboratory): functional simulation of one workgroup (8 waves, two staggered wave groups)
    over three output tiles incl. a ragged last row tile -- result against numpy, LDS-DMA / barrier / counted-wait protocol, static wait-state audit"""
    import numpy as np
    import struct
    from tools.attn_asm.gen_gemm import GemmGen, KERNEL_NAME, KARG_BYTES
    from tools.attn_asm.sim import Machine, bf16_round, bf16_to_f32
    from tools.attn_asm.checks import audit
    M, N, K = 300, 256, 256
    g = GemmGen(K=K, stride=1, sync=sync)          # sync: barriers per K-tile (4: fenced R | M intervals; 1: groups skewed by a k-step)
    prog = g.build()
    assert audit(prog) == []
    rng = np.random.default_rng(0)
    Ab = bf16_round(rng.standard_normal((M, K)).astype(np.float32)).astype(np.uint16)
    Wb = bf16_round((rng.standard_normal((N, K)) * 0.1).astype(np.float32)).astype(np.uint16)
    bias = rng.standard_normal(N).astype(np.float32)
    m = Machine(prog, nwaves=8, lds_bytes=g.lds_bytes)
    pA, pW, pB = m.alloc(Ab.nbytes), m.alloc(Wb.nbytes), m.alloc(bias.nbytes)
    pO = m.alloc(M * N * 2 + 4096)
    m.write(pA, Ab); m.write(pW, Wb); m.write(pB, bias)
    m.write(pO, np.full(M * N + 2048, 0x7FC1, dtype=np.uint16))
    ntiles = (M + 255) // 256 * (N // 256)
    ka = m.alloc(KARG_BYTES)
    m.write(ka, np.frombuffer(struct.pack("<QQQQiiiiIiiiQ", pA, pW, pB, pO, M, N, N // 256, ntiles, 0, 1, ntiles, 0, 0), dtype=np.uint8))

    def setup(w):
        w.s[0], w.s[1], w.s[2] = ka & 0xFFFFFFFF, ka >> 32, 0
        w.v[0] = np.arange(64, dtype=np.uint32) + 64 * w.wid
    m.run(KERNEL_NAME, setup)
    out = m.read(pO, (M * N + 2048) * 2).view(np.uint16)
    got = bf16_to_f32(out[:M * N].astype(np.uint32)).reshape(M, N)
    ref = bf16_to_f32(Ab.astype(np.uint32)).astype(np.float64) @ bf16_to_f32(Wb.astype(np.uint32)).astype(np.float64).T + bias
    assert m.violations == []
    assert np.all(out[M * N:] == 0x7FC1), "rows beyond M were written"
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6          # one bf16 rounding of the result
