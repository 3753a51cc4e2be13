"""GPU: the configuration that has to meet the north-star bar, pinned at FULL size, and the range behaviour of the fp16 residual stream.

* BASELINE configs[1] geometry (DINOv2 ViT-B/14, 518 x 518, 1370 tokens, 12 layers), two images, against the f32 CPU oracle
  (oracle/vit.py: the transformers 5.15 Dinov2 arithmetic of data/utils/feature_extractor.py:49-59, pinned by G8) followed by the f32
  decoder (oracle/decoder.py: models/modules/DBA.py:31-59, pinned by G1/G2): mask logits of the fp16-operand build within 1e-3
  (the north-star tolerance, written here) with either residual-stream type; the bf16 build asserted at its measured level so that
  a regression shows.
* Residual magnitudes of 1e3 and 3e4 (DINOv2 checkpoints are known for a few massive-activation channels): the fp16 stream must follow
  the f32 stream to the operand type's own rounding, must never emit inf / NaN, and must REPORT values it cannot hold (saturation
  counter -> FloatingPointError) instead of passing them on.
* resid="auto" is a property of the engine, not of the batch: an image's key map at batch 1 equals its key map inside a batch.
"""
import json
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from ucod_dpl_amd import ops, native as N  # noqa: E402
from ucod_dpl_amd.vit_engine import ViTEngine  # noqa: E402
from ucod_dpl_amd.data.utils.feature_extractor import random_state_dict, trained_like_state_dict, ARCHS  # noqa: E402
from oracle import decoder as OD, vit as OV  # noqa: E402
from oracle.resize import torch_bilinear  # noqa: E402

DEV = "cuda"
BAR = 1e-3                                                     # BASELINE.json north_star: mask logits within 1e-3 of the reference


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def record(name, values):
    """Measured values of this run, for DESIGN.md / profiles (gpurun_out/ is merged back from the GPU box)."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_c2_measured.jsonl"), "a") as f:
        f.write(json.dumps({"test": name, **values}) + "\n")


def device_logits(key_dev, dec, n, D):
    """The f32 device decoder on a device key map: 1x1 conv on the native grid, bilinear 37 -> 68, pixel-axis norms, fg head."""
    emb = dec["learnable_embedding"].reshape(128).to(DEV)
    hw = torch.cat((dec["conv_out_fg.weight"].reshape(64), dec["conv_out_bg.weight"].reshape(64))).to(DEV)
    hb = torch.cat((dec["conv_out_fg.bias"], dec["conv_out_bg.bias"])).to(DEV)
    w, b = dec["decoupling.weight"].reshape(128, D).to(DEV), dec["decoupling.bias"].to(DEV)
    d = ops.bilinear_resize(ops.dba_project(key_dev, w, b).view(n, 128, *key_dev.shape[-2:]), 68, 68).view(n, 128, 68 * 68)
    return ops.dba_heads(d, 0, emb, ops.dba_colnorm(d, 0, emb), hw, hb, want_bg=False)[0].view(n, 1, 68, 68).cpu()


@pytest.fixture(scope="module")
def c2():
    """Two images of BASELINE configs[1] through the f32 oracle (about 2 s of CPU per image on the GPU box's host)."""
    arch, n = "dinov2_vitb14", 2
    D, heads, L, P, _, _ = ARCHS[arch]
    sd = random_state_dict(arch, 0, 518)
    img = torch.randn(n, 3, 518, 518, generator=torch.Generator().manual_seed(2024))
    with torch.no_grad():
        _, key = OV.dinov2_forward(img, sd, heads=heads, patch=P, eps=1e-6, full_last_layer=False)
        dec = OD.init_params(D, torch.Generator().manual_seed(42))
        fg, _, _ = OD.rev_decoder_forward(torch_bilinear(key, 68, 68), dec, orth="gram")
    return dict(sd=sd, img=img, key=key, fg=fg, dec=dec, heads=heads, D=D, n=n)


@pytest.mark.parametrize("half,resid,logit_tol,key_tol", [
    ("f16", "auto", BAR, 2.5e-3),        # the DEFAULT engine (round 6): fp16 operands on the fp16 stream, LayerNorm folded into QKV / fc1 (measured 6.0e-4 / 1.2e-3)
    ("f16", "f16", BAR, 2.5e-3),         # the same, spelled out
    ("f16", "f32", BAR, 1.5e-3),         # fp16 operands on the f32 residual stream (measured 3.8e-4 / 6.9e-4)
    ("bf16", "auto", 5e-3, 8e-3),        # BASELINE configs[1] dtype: asserted at its measured level (3.2e-3 / 5.5e-3), not at the bar
    ("bf16", "f32", 5e-3, 8e-3),
])
def test_c2_full_size_logits_against_the_oracle(c2, half, resid, logit_tol, key_tol):
    eng = ViTEngine(c2["sd"], heads=c2["heads"], eps=1e-6, device=DEV, half=half, resid=resid)
    assert eng.resid16 == {("f16", "auto"): True, ("f16", "f16"): True, ("f16", "f32"): False, ("bf16", "auto"): True, ("bf16", "f32"): False}[(half, resid)]
    assert eng.ln_fold == (half == "f16" and eng.resid16)
    key_dev = eng(c2["img"].to(DEV))
    eng.check_overflow(wait=True)
    fd = device_logits(key_dev, c2["dec"], c2["n"], c2["D"])
    key_rel, logit_abs = rel_l2(key_dev.cpu(), c2["key"]), float((fd - c2["fg"]).abs().max())
    flipped = float(((fd > 0) != (c2["fg"] > 0)).float().mean())
    record("c2_full_size", dict(half=half, resid=resid, stream="f16" if eng.resid16 else "f32", key_rel_l2=key_rel, logit_max_abs=logit_abs,
                                logit_rel_l2=rel_l2(fd, c2["fg"]), mask_flipped_fraction=flipped))
    assert logit_abs <= logit_tol, (half, resid, logit_abs)
    assert key_rel <= key_tol, (half, resid, key_rel)
    assert flipped <= (0.0 if half == "f16" else 2e-4), flipped          # Delta-MAE of the thresholded masks


@pytest.fixture(scope="module")
def c2_peaked():
    """The same two images through a TRAINED-LIKE synthetic checkpoint (feature_extractor.trained_like_state_dict: query / key weights x 4 ->
    pre-softmax scores of standard deviation ~4 and row entropy ~3.7 against ln 1370 = 7.2; LayerScale in 0.1 .. 1; two massive residual
    channels of magnitude 200): the regime a real DINOv2 checkpoint puts the kernels in, which the trunc-normal init (flat attention
    rows) does not.  VERDICT r3 missing #2 / data/utils/feature_extractor.py:15-29,49-59."""
    arch, n = "dinov2_vitb14", 2
    D, heads, L, P, _, _ = ARCHS[arch]
    sd = trained_like_state_dict(arch, 0, 518)
    img = torch.randn(n, 3, 518, 518, generator=torch.Generator().manual_seed(2024))
    with torch.no_grad():
        _, key = OV.dinov2_forward(img, sd, heads=heads, patch=P, eps=1e-6, full_last_layer=False)
        dec = OD.init_params(D, torch.Generator().manual_seed(42))
        fg, _, _ = OD.rev_decoder_forward(torch_bilinear(key, 68, 68), dec, orth="gram")
        # the REFERENCE's own launcher numerics on these weights: the same oracle with torch-autocast roundings (oracle/vit.py::autocast_rounding,
        # checked against torch.autocast on the HF module in tests/test_oracle_autocast.py) against its f32 self
        ac = {}
        for dt in (torch.float16, torch.bfloat16):
            _, key_ac = OV.dinov2_forward(img, sd, heads=heads, patch=P, eps=1e-6, full_last_layer=False, autocast=dt)
            fg_ac, _, _ = OD.rev_decoder_forward(torch_bilinear(key_ac, 68, 68), dec, orth="gram")
            ac[dt] = dict(logit_max_abs=float((fg_ac - fg).abs().max()), key_rel_l2=rel_l2(key_ac, key))
            record("c2_peaked_reference_autocast", dict(autocast=str(dt), **ac[dt]))
    return dict(sd=sd, img=img, key=key, fg=fg, dec=dec, heads=heads, D=D, n=n, autocast=ac)


# (half, resid, attn_variant): logit max-abs / key relative L2 / flipped-mask fraction asserted at ~2x the measurement
# (profiles/r04_parity_measured.jsonl, rows "c2_peaked"); the fp16-operand rows are ALSO held to the north-star bar where they meet it
PEAKED = {
    # measured (MI355X, round 4; reference logits reach |2.8| here against |0.6| on the flat init):
    ("f16", "f32", 0): (4e-3, 2e-3, 2e-4),           # 2.04e-3 / 1.06e-3 / 0      <- the bar-meeting build of the flat init does NOT meet 1e-3 here
    ("f16", "auto", 0): (7e-3, 3.5e-3, 4.5e-4),      # the default engine = the row below (round 6)
    ("f16", "f16", 0): (7e-3, 3.5e-3, 4.5e-4),       # 3.41e-3 / 1.68e-3 / 0; round 5 (LayerNorm folded into QKV / fc1): 3.53e-3 / 1.62e-3 / 0 .. 2.2e-4 (two pixels)
    ("bf16", "auto", 0): (4e-2, 1.7e-2, 2e-3),       # 1.96e-2 / 8.57e-3 / 7.6e-4
    ("bf16", "f32", 0): (4e-2, 1.7e-2, 2e-3),        # 2.08e-2 / 8.44e-3 / 8.7e-4
    ("bf16", "auto", 8): (0.21, 0.1, 1.4e-2),        # fp8 attention path (BASELINE configs[4]): 1.05e-1 / 5.07e-2 / 6.6e-3 -- 35x the flat-init figure
    ("bf16", "auto", 66): (4e-2, 1.7e-2, 2e-3),      # attn_fwd_v6_kernel inside the engine (the assembly kernels are laboratory code since round 5)
    # round 6 -- the rows HELD TO THE BAR: the split-operand engine (SplitViTEngine: f32 residual stream, every product on two / three bf16 terms per operand)
    ("split2", "f32", 0): (BAR, 5e-5, 0.0),          # 3.5e-5 / 1.6e-5 / 0
    ("split3", "f32", 0): (BAR, 8e-6, 0.0),          # 5.7e-6 / 2.7e-6 / 0   (f32-equivalent: what the feature-cache pass runs)
}


@pytest.mark.parametrize("half,resid,av", list(PEAKED))
def test_c2_full_size_logits_on_trained_like_weights(c2_peaked, half, resid, av):
    c = c2_peaked
    logit_tol, key_tol, flip_tol = PEAKED[(half, resid, av)]
    if half.startswith("split"):
        from ucod_dpl_amd.vit_engine import SplitViTEngine
        eng = SplitViTEngine(c["sd"], heads=c["heads"], eps=1e-6, device=DEV, terms=int(half[-1]))
    else:
        eng = ViTEngine(c["sd"], heads=c["heads"], eps=1e-6, device=DEV, half=half, resid=resid, attn_variant=av)
    key_dev = eng(c["img"].to(DEV))
    eng.check_overflow(wait=True)                              # residual magnitudes of ~200 + updates: in range of the fp16 stream
    assert bool(torch.isfinite(key_dev).all())
    fd = device_logits(key_dev, c["dec"], c["n"], c["D"])
    key_rel, logit_abs = rel_l2(key_dev.cpu(), c["key"]), float((fd - c["fg"]).abs().max())
    flipped = float(((fd > 0) != (c["fg"] > 0)).float().mean())
    record("c2_peaked", dict(half=half, resid=resid, attn_variant=av, stream="f16" if eng.resid16 else "f32", key_rel_l2=key_rel,
                             logit_max_abs=logit_abs, logit_rel_l2=rel_l2(fd, c["fg"]), mask_flipped_fraction=flipped,
                             logit_abs_max_of_reference=float(c["fg"].abs().max())))
    assert logit_abs <= logit_tol, (half, resid, av, logit_abs)
    assert key_rel <= key_tol, (half, resid, av, key_rel)
    assert flipped <= flip_tol, (half, resid, av, flipped)
    # ... and no further from f32 than the reference's own autocast forward in the same operand type is (fp16: what its launcher runs, scripts/
    # launch_train_first_stage.sh:20; measured on one image: 2.95e-3 fp16, 2.3e-2 bf16).  The fp16 residual stream adds its own rounding: 1.6x.
    if half.startswith("split"):
        assert logit_abs <= BAR                                  # (spelled out: these are the configurations that meet the north-star bar on these weights)
    elif av != 8:
        ref_dev = c["autocast"][torch.float16 if half == "f16" else torch.bfloat16]["logit_max_abs"]
        assert logit_abs <= (1.6 if (half == "f16" and eng.resid16) else 1.25) * ref_dev, (half, resid, av, logit_abs, ref_dev)


def test_c5_fp8_attention_path_full_depth_against_the_oracle(c2):
    """BASELINE configs[4]: ViTEngine(attn_variant=8) -- Q, K, V and the probabilities in OCP e4m3 on v_mfma_scale_f32_32x32x64_f8f6f4,
    everything else bf16 -- at 518 x 518, full depth, against the f32 oracle.  A throughput-only configuration: what it costs in
    accuracy is MEASURED here and asserted at that level (key relative L2, logit max-abs, fraction of mask pixels on the other side of the
    threshold), next to the bf16 engine on the same images."""
    e8 = ViTEngine(c2["sd"], heads=c2["heads"], eps=1e-6, device=DEV, attn_variant=8, half="bf16")
    eb = ViTEngine(c2["sd"], heads=c2["heads"], eps=1e-6, device=DEV, half="bf16")
    img = c2["img"].to(DEV)
    k8, kb = e8(img), eb(img)
    f8, fb = device_logits(k8, c2["dec"], c2["n"], c2["D"]), device_logits(kb, c2["dec"], c2["n"], c2["D"])
    m = dict(fp8_key_rel_l2=rel_l2(k8.cpu(), c2["key"]), bf16_key_rel_l2=rel_l2(kb.cpu(), c2["key"]),
             fp8_logit_max_abs=float((f8 - c2["fg"]).abs().max()), bf16_logit_max_abs=float((fb - c2["fg"]).abs().max()),
             fp8_logit_rel_l2=rel_l2(f8, c2["fg"]), fp8_mask_flipped_fraction=float(((f8 > 0) != (c2["fg"] > 0)).float().mean()),
             bf16_mask_flipped_fraction=float(((fb > 0) != (c2["fg"] > 0)).float().mean()))
    record("c5_fp8_full_depth", m)
    assert bool(torch.isfinite(k8).all())
    # Measured (MI355X, round 3, profiles/r03_parity_measured.jsonl): key relative L2 5.74e-3 against the bf16 engine's 5.55e-3, logit max-abs
    # 2.95e-3 against 2.99e-3, no mask pixel flipped.  With random-init weights the attention rows are nearly flat (scores of a few
    # tenths), so the e4m3 rounding of the scores averages out over 1370 keys; a trained checkpoint's peaked rows sit closer to the
    # kernel-level figure (7 % on unit-Gaussian q, k: tests/test_gpu_fp8_attention.py).  Asserted at 2x the measurement.
    assert m["fp8_key_rel_l2"] < 1.2e-2, m
    assert m["fp8_logit_max_abs"] < 6e-3, m
    assert m["fp8_mask_flipped_fraction"] < 1e-3, m


def test_c2_key_map_does_not_depend_on_the_batch(c2):
    """resid="auto" picks the stream type from the engine, never from the batch size: image 0 alone (small-tile kernels) and image 0 inside a
    batch of 6 (large-tile kernels) go through the same arithmetic types.  The two passes differ only in the f32 summation order of the two
    tile shapes; twelve layers of re-rounding to the 16-bit operand type turn those last-bit differences into the operand type's own
    noise level (bf16: measured 4.0e-3, the same size as the bf16 engine's distance from the oracle; fp16: 8x finer), never more."""
    img = torch.cat((c2["img"], torch.randn(4, 3, 518, 518, generator=torch.Generator().manual_seed(5))), 0).to(DEV)
    for half in ("bf16", "f16"):
        eng = ViTEngine(c2["sd"], heads=c2["heads"], eps=1e-6, device=DEV, half=half)
        assert eng._desc(1, 518, 518).resid16 == eng._desc(6, 518, 518).resid16 == 1          # (round 6: both defaults run the fp16 stream; fp16 operands fold LayerNorm)
        assert eng._desc(1, 518, 518).ln_fold == eng._desc(6, 518, 518).ln_fold == int(half == "f16")
        k6 = eng(img).clone()
        k1 = eng(img[:1].contiguous())
        r = rel_l2(k1, k6[:1])
        record("batch_independence", dict(half=half, rel_l2_batch1_vs_batch6=r))
        assert r < (8e-3 if half == "bf16" else 1.5e-3), (half, r)
        eng.check_overflow(wait=True)


# ------------------------------------------------------------------------------------------------ residual magnitudes of 1e3 .. 3e4
def _massive_state_dict(mag, seed=3):
    """A small DINOv2-shaped model (D = 256, 4 heads, 6 layers, 224 x 224 -> 257 tokens) with "massive activations": the position embedding
    puts `mag` into two channels of the CLS token and of a few patch tokens, and LayerScale = 1 lets every layer add O(1) on top -- the
    residual stream then carries values of magnitude `mag` from the first layer to the last, as DINOv2 checkpoints do."""
    ARCHS["massive_vit"] = (256, 4, 6, 14, 224, True)
    sd = random_state_dict("massive_vit", seed=seed)
    pos = sd["embeddings.position_embeddings"]
    for t in (0, 17, 100, 256):
        pos[0, t, 5] = mag
        pos[0, t, 200] = -0.75 * mag
    return sd


@pytest.mark.parametrize("half", ["bf16", "f16"])
@pytest.mark.parametrize("mag", [1.0e3, 3.0e4])
def test_fp16_stream_follows_the_f32_stream_at_large_residuals(half, mag):
    sd = _massive_state_dict(mag)
    img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        _, ref = OV.dinov2_forward(img, sd, heads=4, patch=14, eps=1e-6, full_last_layer=False)
    e32 = ViTEngine(sd, heads=4, eps=1e-6, device=DEV, half=half, resid="f32")
    e16 = ViTEngine(sd, heads=4, eps=1e-6, device=DEV, half=half, resid="f16")
    k32, k16 = e32(img.to(DEV)).cpu(), e16(img.to(DEV)).cpu()
    e16.check_overflow(wait=True)                               # in range: nothing saturated
    assert bool(torch.isfinite(k16).all()) and bool(torch.isfinite(k32).all())
    d32, d16, between = rel_l2(k32, ref), rel_l2(k16, ref), rel_l2(k16, k32)
    record("massive_residual", dict(half=half, mag=mag, f32_stream_vs_oracle=d32, f16_stream_vs_oracle=d16, f16_vs_f32_stream=between))
    # the fp16 stream may not cost more than the operand type's own rounding already does: within 1.5x of the f32-stream engine's distance
    # from the oracle (plus the fp16 rounding of one layer's update), and the two engines agree to that level
    floor = 2.0 ** -11
    assert d16 <= 1.5 * d32 + 2 * floor, (half, mag, d16, d32)
    assert between <= 1.5 * d32 + 2 * floor, (half, mag, between, d32)


@pytest.mark.parametrize("half", ["bf16", "f16"])
def test_fp16_stream_saturates_and_reports(half):
    """A residual value fp16 cannot hold (1e5): the stream clamps to +-65504 (no inf, no NaN key map) and the engine raises."""
    sd = _massive_state_dict(1.0e5)
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(10))
    e16 = ViTEngine(sd, heads=4, eps=1e-6, device=DEV, half=half, resid="f16")
    k16 = e16(img.to(DEV))
    assert bool(torch.isfinite(k16).all())
    with pytest.raises(FloatingPointError):
        e16.check_overflow(wait=True)
    e16.check_overflow(wait=True)                               # the counter was reset by the report
    k_again = e16(img.to(DEV))
    with pytest.raises(FloatingPointError):                     # the NEXT pass (non-blocking poll inside forward, then the explicit wait) reports again
        e16.check_overflow(wait=True)
    assert torch.equal(k_again, k16)
    # the f32 stream holds the same model without complaint
    e32 = ViTEngine(sd, heads=4, eps=1e-6, device=DEV, half=half, resid="f32")
    assert bool(torch.isfinite(e32(img.to(DEV))).all())


def test_saturation_is_reported_by_the_engine_that_caused_it():
    """Two engines with the fp16 stream on one GPU: each counts into its own device word (ucod_resid16_overflow_bind), so the healthy one never raises
    for the other's saturation, in either order of polling (the counter used to be one word per device)."""
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(11))
    bad = ViTEngine(_massive_state_dict(1.0e5), heads=4, eps=1e-6, device=DEV, half="bf16", resid="f16")
    good = ViTEngine(_massive_state_dict(1.0e3), heads=4, eps=1e-6, device=DEV, half="bf16", resid="f16")
    for order in ((bad, good), (good, bad)):
        for e in order:
            e(img.to(DEV))
        good.check_overflow(wait=True)                          # no FloatingPointError
        good(img.to(DEV))                                       # (the non-blocking poll inside forward) neither
        with pytest.raises(FloatingPointError):
            bad.check_overflow(wait=True)
    # the per-device word is untouched by bound passes
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    N.check(N.load().ucod_resid16_overflow_fetch(host.data_ptr(), N.stream()), "fetch")
    torch.cuda.synchronize()
    assert int(host[0]) == 0


def test_backward_engine_on_both_residual_streams():
    """Round 4: the backbone-backward engine saves the fp16 residual stream by default (bf16 operands), like the frozen engine; LayerNorm backward reads
    it (ucod_layernorm_bwd_ex).  Key map and LoRA gradients of the two streams agree at the level of the operands' own rounding."""
    from ucod_dpl_amd.vit_engine import ViTLoRAEngine
    ARCHS["tiny_vit"] = (128, 2, 3, 14, 70, True)
    sd = random_state_dict("tiny_vit", seed=1)
    out = {}
    for resid in ("auto", "f32"):
        eng = ViTLoRAEngine(sd, heads=2, device=DEV, resid=resid, generator=torch.Generator().manual_seed(3))
        lsd = eng.lora_state_dict()
        g = torch.Generator().manual_seed(5)
        for k in lsd:
            if "lora_B" in k:
                lsd[k] = 0.05 * torch.randn(lsd[k].shape, generator=g)
        eng.load_lora_state_dict(lsd)
        assert eng.resid16 is (resid == "auto") and eng._desc(64, 70, 70).resid16 == int(resid == "auto")
        x = torch.randn(4, 3, 70, 70, generator=torch.Generator().manual_seed(6))
        key = eng.forward_train(x.to(DEV))
        eng.backward(torch.randn(key.shape, generator=torch.Generator().manual_seed(7)).to(DEV))
        eng.check_overflow(wait=True)
        out[resid] = (key.cpu(), eng.lora_grad.cpu().clone())
    assert rel_l2(out["auto"][0], out["f32"][0]) < 5e-3
    assert rel_l2(out["auto"][1], out["f32"][1]) < 2e-2
    record("lora_engine_streams", dict(key_rel_l2=rel_l2(out["auto"][0], out["f32"][0]), grad_rel_l2=rel_l2(out["auto"][1], out["f32"][1])))


def test_backward_engine_reports_saturation_of_its_fp16_stream():
    """The training pass on the fp16 residual stream saturates and counts like the frozen pass: forward_train + check_overflow raises for a model whose
    residual stream reaches 1e5, not for one at 1e3; resid="f32" holds the former without complaint."""
    from ucod_dpl_amd.vit_engine import ViTLoRAEngine
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(12)).to(DEV)
    ok = ViTLoRAEngine(_massive_state_dict(1.0e3), heads=4, device=DEV)
    ok.forward_train(img)
    ok.check_overflow(wait=True)
    bad = ViTLoRAEngine(_massive_state_dict(1.0e5), heads=4, device=DEV)
    key = bad.forward_train(img)
    assert bool(torch.isfinite(key).all())
    with pytest.raises(FloatingPointError):
        bad.check_overflow(wait=True)
    f32 = ViTLoRAEngine(_massive_state_dict(1.0e5), heads=4, device=DEV, resid="f32")
    assert bool(torch.isfinite(f32.forward_train(img)).all())
    f32.check_overflow(wait=True)

