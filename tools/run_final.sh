# Round-end evidence (run on the MI355X box from the repo root): bash tools/run_final.sh <label>
S=${1:-r02f}
python bench.py > gpurun_out/bench_$S.json 2> gpurun_out/bench_$S.err; tail -2 gpurun_out/bench_$S.err
bash tools/refresh_evidence.sh $S > gpurun_out/refresh_$S.log 2>&1
python bench.py --arch dinov2_vitl14 --batch 16 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_${S}_c4_vitl14_b16.json 2>/dev/null
python bench.py --arch dino_vits8 --image 224 --batch 2 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_${S}_c1_vits8_b2.json 2>/dev/null
python bench.py --batch 64 --attn-variant 8 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_${S}_c5_fp8.json 2>/dev/null
python bench.py --batch 64 --lora-steps 0 --no-cpu-baseline > gpurun_out/bench_${S}_c5_geometry_bf16.json 2>/dev/null
python - <<PY
import json
for f in ("bench_$S", "bench_${S}_c4_vitl14_b16", "bench_${S}_c1_vits8_b2", "bench_${S}_c5_fp8", "bench_${S}_c5_geometry_bf16"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("hbm_row", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
d = json.load(open("gpurun_out/bench_$S.json"))
print(json.dumps(d["cpu_baseline"])[:1500])
print(d["f16_operands_option"], d["backbone_backward_mode"]["value"], d["discriminator_phase"]["value"])
PY
