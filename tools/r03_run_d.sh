#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/parity_c2_measured.jsonl gpurun_out/fp8_attention_measured.jsonl
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03d_pytest.txt 2>&1
tail -8 gpurun_out/r03d_pytest.txt
timeout 900 python bench.py > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err
tail -c 400 gpurun_out/r03d_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03d_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "logit_max_abs", "bar_met", "host_enqueue_ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["frac"])
print("  lora:", {k: v for k, v in d["backbone_backward_mode"].items() if k != "what"})
print("  kernels:", {k: (v.get("avg_us"), v.get("tflops")) for k, v in d["kernels"].items() if "gemm" in k or "attention" in k or k == "layernorm"})
PY
timeout 1500 bash tools/r03_evidence.sh > gpurun_out/r03d_evidence.log 2>&1
tail -90 gpurun_out/r03d_evidence.log
