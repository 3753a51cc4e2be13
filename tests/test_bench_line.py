"""CPU: the committed bench line of the round (profiles/bench_r05d.json, written by `python bench.py` on an MI355X) keeps the driver's contract --
the keys, their types, and the arithmetic between them (value = images of a step / ms_per_step, roofline.frac = achieved / peak, metric / config of
BASELINE.json configs[1]).  bench.py itself runs in the GPU suite (tests/test_gpu_multirank.py: 2 and 8 ranks started by the script)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name="bench_r05d.json"):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_contract_keys_and_arithmetic():
    d = _line()
    for k, ty in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                  ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], ty), (k, type(d[k]))
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["global_batch"]
    assert abs(d["value"] - B / d["ms_per_step"] * 1e3) / d["value"] < 2e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == d["unit"] and c["cores"] >= 1


def test_line_is_on_the_baseline_configuration():
    d = _line()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        b = json.load(f)
    assert "images/sec" in d["metric"] and "images" in b["metric"].lower()
    assert "configs[1]" in d["config"]["workload"] and d["config"]["global_batch"] == 32
    # the headline configuration is the one that meets the north-star logit bar, and says so
    assert d["bar_met"] is True and d["logit_max_abs"] <= d["bar"] == 1e-3 and d["value_at_bar"] == d["value"]
