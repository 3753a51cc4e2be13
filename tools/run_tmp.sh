for av in 2 7 2 7; do
python bench.py --steps 20 --warmup 5 --lora-steps -1 --no-cpu-baseline --attn-variant $av > gpurun_out/b_av.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/b_av.json")); print("attn variant $av", d["value"], d["ms_per_step"], d["kernels"]["attention_fwd"]["avg_us"])
PY
done
