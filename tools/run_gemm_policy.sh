python tools/gemm_order_sweep.py --mode policy --shapes qkv,fc1 --rounds 7 > gpurun_out/sweep_policy.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | tail -3
for cfg in "0 0" "0 2" "0 16" "1 0" "1 2"; do set -- $cfg
UCOD_GEMM_PREFETCH=$1 UCOD_GEMM_ST_AUX=$2 python bench.py --steps 20 --warmup 5 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_pf$1_aux$2.json 2> gpurun_out/bench_pf$1_aux$2.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_pf$1_aux$2.json"))
k=d["kernels"]
print("pf=$1 aux=$2", d["value"], d["roofline"]["serial_ms_per_step_without_events"], {n:k[n]["avg_us"] for n in ("gemm_bf16_qkv_bias","gemm_bf16_fc1_gelu","gemm_bf16_proj_fc2_scale_resid","attention_fwd","layernorm")})
PY
done
UCOD_GEMM_NO_MIXED=1 python bench.py --steps 20 --warmup 5 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_nomixed.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/bench_nomixed.json"))
k=d["kernels"]
print("nomixed", d["value"], d["roofline"]["serial_ms_per_step_without_events"], {n:k[n]["avg_us"] for n in ("gemm_bf16_qkv_bias","gemm_bf16_fc1_gelu","gemm_bf16_proj_fc2_scale_resid","attention_fwd","layernorm")})
PY
