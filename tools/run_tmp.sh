python -m pytest tests/test_gpu_fp8_attention.py -x -q 2>&1 | grep -E "^E|passed|failed|assert" | head -20
for av in 2 8; do
python bench.py --steps 10 --warmup 3 --batch 64 --attn-variant $av --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_b64_av$av.json 2> gpurun_out/bench_b64_av$av.err || tail -3 gpurun_out/bench_b64_av$av.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_b64_av$av.json")); k=d["kernels"]
print("av=$av", d["config"]["workload"][:60], d["value"], d["roofline"]["serial_ms_per_step_without_events"], {n:k[n]["avg_us"] for n in ("gemm_bf16_qkv_bias","attention_fwd","layernorm")})
PY
done
