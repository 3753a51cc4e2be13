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


@pytest.mark.parametrize("fname,text", [("attn_fwd_pw64_bf16.s", lambda: gen_attn.kernel_text("bf16")[0]),
                                        ("attn_fwd_pw32_bf16.s", lambda: gen_attn32.kernel_text("bf16")[0])])
def test_committed_assembly_is_current(fname, text):
    with open(os.path.join(ASM, fname)) as f:
        committed = f.read().split("\n", 1)[1]          # first line: the GENERATED banner
    assert committed == text(), f"{fname} is stale: run `make -C ucod_dpl_amd/csrc variants`"
