"""CPU: bench.py's line builder on CANNED measurements (the previous form asserted on a committed JSON artefact and could not fail when
bench.py broke).  `bench.build_line(args, measurements)` is pure host arithmetic: the driver's contract (keys, types, value = images of a step / time,
roofline.frac = achieved / peak from the algorithmic flops and the per-class HIP-event totals, metric / workload of BASELINE.json configs[1]), the parity-bar
fields, the sustained-run rule (a burst the chip does not hold is not the headline) and the split-operand configuration's pricing.  The committed line of the
round is checked against the same contract by `check_line`.  bench.py itself runs in the GPU suite (tests/test_gpu_multirank.py: 2 and 8 ranks started by the script)."""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def args(**kw):
    base = dict(gpus=1, steps=20, warmup=8, batch=32, arch="dinov2_vitb14", image=518, full_last_layer=False, attn_variant=2, streams=2, no_pipeline=False, half="f16",
                resid="auto", ln_fold="auto", sustain_s=40.0)
    base.update(kw)
    return argparse.Namespace(**base)


def canned(**kw):
    """A ViT-B/14 batch-32 step as an MI355X shows it (per-class totals over 20 steps of the serial pass, ms): QKV 11 x 154 us, fc1 11 x 222 us, proj + fc2 22 x 122.5 us,
    attention 11 x 212 us, one LayerNorm of 25 us, one statistics launch, patch embedding, key hook, the decoder step's small kernels."""
    steps = 20
    per_step = [("gemm_bf16_qkv_bias", 11, 154.0), ("gemm_bf16_fc1_gelu", 11, 222.0), ("gemm_bf16_proj_fc2_scale_resid", 22, 122.5), ("attention_fwd", 11, 212.0),
                ("layernorm", 1, 25.0), ("row_stats", 1, 13.3), ("gemm_bf16_patch_embed", 1, 60.0), ("gemm_bf16_key_nchw", 1, 40.0), ("dba_project_f32", 1, 30.0), ("adamw_ema", 1, 5.0)]
    m = dict(world=1, B=32, D=768, heads=12, L=12, P=14, kpad=640, image=518, resid16=True, ln_fold=True, dt=steps * 9.364e-3, dt_serial=steps * 10.1e-3, dt_serial_plain=steps * 10.0e-3,
             classes=[(n, steps * c * us * 1e-3, steps * c) for n, c, us in per_step], final_loss=1.234567, dis_phase={"value": 1.0}, lora_mode=None, host_enqueue=steps * 3.1e-3,
             host_threads=128, pinned_cores=128, ranks_seen=[[0, 0, 0, "0000:05:00"]], cpu=None, others={}, traffic={"gemm_bf16_proj_fc2_scale_resid": {"traffic_bytes": 3.7e8}},
             traffic_source="canned", sustained=None, collectives="none (world size 1: short-circuit)")
    m.update(kw)
    return m


def check_line(d, n_gpus=1):
    for k, ty in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                  ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], ty), (k, type(d[k]))
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == n_gpus
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    if d.get("cpu_baseline"):
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["cores"] >= 1


def test_contract_keys_and_arithmetic_of_the_built_line():
    a, m = args(), canned()
    d = json.loads(json.dumps(bench.build_line(a, m)))                # (the line must survive JSON)
    check_line(d)
    assert abs(d["value"] - 32 / 9.364e-3) < 0.5 and abs(d["ms_per_step"] - 9.364) < 1e-3 and d["value_timed_region"] == d["value"] and d["value_is"] == "the timed region"
    # the dominant class by time in the step is proj + fc2 (22 x 122.5 us); its rate from the ALGORITHMIC flops of the 11 layers that run whole
    r = d["roofline"]
    flops = 11 * (2.0 * 32 * 1370 * 768 * 768 + 2.0 * 32 * 1370 * 768 * 3072)
    assert r["kernel"] == "gemm_bf16_proj_fc2_scale_resid" and abs(r["achieved"] - flops / (22 * 122.5e-6) / 1e12) < 0.2 and r["peak"] == 2500.0
    assert abs(r["frac"] - 0.4222) < 2e-3 and r["traffic"] == 3.7e8 and r["traffic_source"] == "canned" and abs(r["avg_launch_us"] - 122.5) < 1e-6
    k = d["kernels"]
    assert abs(k["gemm_bf16_qkv_bias"]["tflops"] - 2.0 * 43840 * 2304 * 768 / 154e-6 / 1e12) < 0.2
    assert abs(k["attention_fwd"]["tflops"] - 4.0 * 32 * 12 * 1370 * 1370 * 64 / 212e-6 / 1e12) < 0.2
    assert abs(r["attention_row"]["frac"] - k["attention_fwd"]["tflops"] / 2500.0) < 1e-3
    assert abs(r["hbm_row"]["achieved"] - 43840 * 768 * 4 / 25e-6 / 1e9) < 1.0 and r["layernorm_launches_per_step"] == 1.0
    assert "configs[1]" in d["config"]["workload"] and d["config"]["global_batch"] == 32 and d["config"]["ln_fold"] is True and d["dtype"] == "f16"
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        b = json.load(f)
    assert "images/sec" in d["metric"] and "images" in b["metric"].lower()
    assert d["configurations"]["f16_f16_stream"]["this_line"] is True and "default" in d["drop_in_default_engine"]


def test_parity_bar_fields_and_the_configuration_table():
    par = lambda l, k: {"logit_max_abs": l, "logit_rel_l2": l / 3, "key_rel_l2": k, "mask_flipped_fraction": 0.0}  # noqa: E731
    cpu = {"value": 0.2263, "unit": "images/s", "cores": 128, "kind": "port", "sample": "4 images",
           "parity_full_size": {"f16_operands_f16_stream": par(6.0e-4, 1.2e-3), "f16_operands": par(3.8e-4, 6.9e-4), "bf16": par(3.2e-3, 5.5e-3), "split2": par(2.5e-5, 1.2e-5),
                                "split3": par(4.0e-6, 2.0e-6), "reference_fp16_autocast_emulation": par(9e-4, 1e-3),
                                "trained_like_weights": {"f16_operands_f16_stream": par(2.9e-3, 1.6e-3), "f16_operands": par(2.0e-3, 1.1e-3), "bf16": par(2.0e-2, 8.6e-3),
                                                         "split2": par(3.5e-5, 1.6e-5), "split3": par(5.7e-6, 2.7e-6), "reference_fp16_autocast_emulation": par(3.1e-3, 2e-3)}}}
    child = lambda v, ms, rs, fold: {"value": v, "unit": "images/s", "ms_per_step": ms, "steps": 20, "dtype": "x", "residual_stream": rs, "ln_fold": fold,  # noqa: E731
                                     "serial_ms_per_step": ms * 1.07, "kernels_avg_us": {"gemm_bf16_qkv_bias": 150.0}, "roofline": {"frac": 0.4}, "how": "child"}
    others = {"f16_f32_stream": child(3200.0, 10.0, "f32", False), "bf16": child(3390.0, 9.44, "fp16", False), "split2": child(1100.0, 29.1, "f32", False),
              "split3": child(600.0, 53.3, "f32", False)}
    d = bench.build_line(args(), canned(cpu=cpu, others=others))
    check_line(d)
    assert d["bar_met"] is True and d["logit_max_abs"] == 6.0e-4 and d["bar"] == 1e-3 and d["bar_met_trained_like_weights"] is False and d["logit_max_abs_trained_like_weights"] == 2.9e-3
    # fastest configuration under the bar on the flat init = this line's; the fastest that ALSO meets it on the trained-like weights = the two-term split pass
    assert d["value_at_bar"] == d["value"] and d["value_at_bar_met_on_trained_like_weights"] is False
    assert d["value_at_bar_on_trained_like_weights"] == 1100.0 and "SplitViTEngine(terms=2)" in d["value_at_bar_on_trained_like_weights_config"]
    c = d["configurations"]
    assert set(c) == {"f16_f16_stream", "f16_f32_stream", "bf16", "split2", "split3"}
    assert c["bf16"]["bar_met"] is False and c["split3"]["bar_met_on_trained_like_weights"] is True and c["f16_f32_stream"]["bar_met_on_trained_like_weights"] is False
    assert d["logit_max_abs_of_reference_fp16_autocast"] == [9e-4, 3.1e-3]


def test_a_burst_the_chip_does_not_hold_is_not_the_headline():
    win = lambda v, mhz: {"steps": 1000, "seconds": 10.0, "images_per_s": v, "held_clock_mhz": mhz}  # noqa: E731
    held = {"seconds": 40.0, "windows": [win(3400.0, 1810.0), win(3390.0, 1800.0), win(3385.0, 1795.0), win(3380.0, 1795.0)], "value_sustained": 3380.0, "held_clock_mhz": 1800.0, "what": "canned"}
    d = bench.build_line(args(), canned(sustained=held))
    assert d["value"] == d["value_timed_region"] and d["value_sustained"] == 3380.0            # within 3 % of the timed region: the timed region stays the value
    r = d["roofline"]
    assert r["held_clock_mhz"] == 1800.0 and abs(r["peak_at_held_clock"] - 2500.0 * 1800 / 2400) < 0.1 and abs(r["frac_at_held_clock"] - r["achieved"] / 1875.0) < 1e-3
    sag = dict(held, windows=[win(3300.0, 1700.0), win(3100.0, 1600.0)], value_sustained=3100.0, held_clock_mhz=1650.0)
    d2 = bench.build_line(args(), canned(sustained=sag))
    assert d2["value"] == 3100.0 and d2["value_timed_region"] > 3400 and "LAST" in d2["value_is"]
    check_line(d2)


def test_split_configuration_is_priced_on_algorithmic_flops():
    steps = 5
    per_step = [("gemm_bf16_bias_f32", 22, 520.0), ("gemm_bf16_proj_fc2_scale_resid", 22, 380.0), ("attention_split_fwd", 11, 900.0), ("layernorm_split", 23, 60.0),
                ("split_operands", 22, 80.0), ("gemm_bf16_patch_embed", 1, 150.0), ("gemm_bf16_key_nchw", 1, 90.0)]
    m = canned(resid16=False, ln_fold=False, dt=steps * 33e-3, dt_serial=steps * 34e-3, dt_serial_plain=steps * 33.8e-3, classes=[(n, steps * c * us * 1e-3, steps * c) for n, c, us in per_step],
               traffic=None, traffic_source=None)
    d = bench.build_line(args(half="split2", steps=steps), m)
    check_line(d)
    assert d["dtype"] == "bf16x2" and d["config"]["residual_stream"] == "f32" and "two-term split" in d["config"]["workload"]
    k, r = d["kernels"], d["roofline"]
    alg = 11 * (2.0 * 43840 * 2304 * 768 + 2.0 * 43840 * 3072 * 768) / (22 * 520e-6) / 1e12
    assert abs(k["gemm_bf16_bias_f32"]["tflops"] - alg) < 0.2 and abs(k["gemm_bf16_bias_f32"]["mfma_issued_tflops"] - 3 * alg) < 0.5
    assert r["kernel"] == "gemm_bf16_bias_f32" and abs(r["mfma_issued_frac"] - 3 * r["frac"]) < 2e-3 and r["traffic"] is None and r["traffic_source"] is None
    assert r["hbm_row"]["kernel"] == "layernorm_split" and abs(r["hbm_row"]["achieved"] - 43840 * 768 * (4 + 6) / 60e-6 / 1e9) < 1.0
    assert abs(r["attention_row"]["achieved"] - 4.0 * 32 * 12 * 1370 * 1370 * 64 / 900e-6 / 1e12) < 0.2


def test_the_committed_line_of_the_round_keeps_the_contract():
    """The driver line the round commits (profiles/bench_r06*.json when it exists, else the last earlier round's) against the same contract."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r06*.json"))) or [os.path.join(ROOT, "profiles", "bench_r05d.json")]
    for path in files:
        with open(path) as f:
            d = json.loads(f.read().strip().splitlines()[-1])
        if "look-twice" in d.get("metric", "").lower() or "Look-Twice" in d.get("metric", ""):
            continue
        check_line(d, n_gpus=d["n_gpus"])
        assert abs(d.get("value_timed_region", d["value"]) - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) / d["value"] < 2e-3
