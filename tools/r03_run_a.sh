#!/bin/bash
# round-3 GPU run A: A/B of the attention forms, the whole GPU test suite (laboratory variants included), one bench line
mkdir -p gpurun_out
python tools/attn_ab.py r2=lab:102 f1=ucod_dpl_amd/_native/libucod_dpl_forms.so:21 f2=ucod_dpl_amd/_native/libucod_dpl_forms.so:22 f3=ucod_dpl_amd/_native/libucod_dpl_forms.so:23 f4=ucod_dpl_amd/_native/libucod_dpl_forms.so:24 product=product:2 > gpurun_out/r03a_attn_ab.txt 2>&1
tail -8 gpurun_out/r03a_attn_ab.txt
rm -f gpurun_out/parity_c2_measured.jsonl
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03a_pytest.txt 2>&1
tail -40 gpurun_out/r03a_pytest.txt
timeout 600 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
tail -c 1500 gpurun_out/r03a_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03a_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "logit_max_abs", "bar_met", "host_enqueue_ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["frac"], d.get("bar_meeting_config"))
    print({k: (v.get("avg_us"), v.get("tflops")) for k, v in d["kernels"].items() if "gemm" in k or "attention" in k or k == "layernorm"})
except Exception as e:
    print("bench parse failed", e)
PY
