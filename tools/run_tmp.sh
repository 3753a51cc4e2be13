python bench.py --steps 20 --warmup 5 --half f16 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err; tail -5 gpurun_out/bench_f16.err
python bench.py --steps 20 --warmup 5 --lora-steps -1 --no-cpu-baseline > gpurun_out/bench_bf16.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_f16","bench_bf16"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); k=d["kernels"]
        print(f, d["value"], d["roofline"]["serial_ms_per_step_without_events"], {n:k[n]["avg_us"] for n in k if k[n]["ms_per_step"]>0.05})
    except Exception as e: print(f, e)
PY
