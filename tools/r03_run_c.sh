#!/bin/bash
mkdir -p gpurun_out
python tools/gemm_ab.py noearly=ucod_dpl_amd/_native/libucod_dpl_noearly.so > gpurun_out/r03c_gemm_ab.txt 2>&1
cat gpurun_out/r03c_gemm_ab.txt | tail -14
python tools/attn_bwd_ab.py > gpurun_out/r03c_attn_bwd_ab.txt 2>&1
tail -4 gpurun_out/r03c_attn_bwd_ab.txt
rm -f gpurun_out/parity_c2_measured.jsonl gpurun_out/fp8_attention_measured.jsonl
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03c_pytest.txt 2>&1
tail -12 gpurun_out/r03c_pytest.txt
grep c5_fp8 gpurun_out/parity_c2_measured.jsonl; cat gpurun_out/fp8_attention_measured.jsonl
timeout 900 python bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
tail -c 600 gpurun_out/r03c_bench.err
timeout 900 python bench.py --batch 64 --attn-variant 8 --lora-steps 0 > gpurun_out/r03c_bench_c5_fp8.json 2> gpurun_out/r03c_bench_c5.err
tail -c 600 gpurun_out/r03c_bench_c5.err
python - <<'PY'
import json
for f in ("gpurun_out/r03c_bench.json", "gpurun_out/r03c_bench_c5_fp8.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, {k: d.get(k) for k in ("value", "ms_per_step", "logit_max_abs", "bar_met", "host_enqueue_ms_per_step")}, d["roofline"]["kernel"], d["roofline"]["frac"])
        print("  bar_meeting:", {k: v for k, v in (d.get("bar_meeting_config") or {}).items() if k != "all_f16_configurations" and k != "what"})
        for c in (d.get("bar_meeting_config") or {}).get("all_f16_configurations", []): print("   ", c)
        print("  kernels:", {k: (v.get("avg_us"), v.get("tflops")) for k, v in d["kernels"].items() if "gemm" in k or "attention" in k or k == "layernorm"})
        print("  lora:", d.get("backbone_backward_mode"))
        print("  f16 vs bf16:", (d.get("f16_vs_bf16_per_kernel") or {}).get("avg_us_bf16_vs_f16"))
        p = (d.get("cpu_baseline") or {}).get("parity_full_size") or {}
        print("  parity:", {k: p.get(k) for k in ("key_rel_l2", "logit_max_abs", "mask_flipped_fraction")})
    except Exception as e:
        print(f, "parse failed", e)
PY
