python bench.py > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err; tail -2 gpurun_out/bench_r02c.err
bash tools/refresh_evidence.sh r02c > gpurun_out/refresh_r02c.log 2>&1; tail -5 gpurun_out/refresh_r02c.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r02c.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel"], d["f16_operands_option"], d["backbone_backward_mode"])
PY
