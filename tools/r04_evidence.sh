#!/bin/bash
# round-4 evidence on the GPU box (from the repo root): rocprofv3 kernel stats (serial, default, LoRA mode with dropout 0.05), PMC traffic,
# launches per timed step, the side configurations' bench lines (with a 1-image parity block), the assembly attention kernel in the step.
# Summaries land in gpurun_out/; the ones to keep are copied into profiles/ by hand.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/refresh_evidence.sh r04 > $O/r04_refresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04_lora -- python3 $R/tools/lora_bench.py 32 2 1 0.05 > $O/prof_r04_lora_bench.txt 2> $O/prof_r04_lora.err
find $O/prof_r04_lora -name '*kernel_trace.csv' -delete
for n in 2 12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/calls_$n -- python3 $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 > $O/calls_$n.json 2> $O/calls_$n.err
  find $O/calls_$n -name '*kernel_trace.csv' -delete
done
python3 $R/tools/per_step_calls.py $O/calls_2 2 $O/calls_12 12 -v > $O/r04_per_step_calls.txt
cd $R
python bench.py --arch dinov2_vitl14 --batch 16 --lora-steps 0 --cpu-images 1 > $O/bench_r04_c4_vitl14_b16.json 2> $O/bench_r04_c4.err
python bench.py --batch 64 --attn-variant 8 --lora-steps 0 --cpu-images 1 > $O/bench_r04_c5_fp8.json 2> $O/bench_r04_c5.err
python bench.py --arch dino_vits8 --image 224 --batch 2 --lora-steps 0 --cpu-images 1 > $O/bench_r04_c1_vits8_b2.json 2> $O/bench_r04_c1.err
UCOD_ATTN_ASM=1 python bench.py --lora-steps 0 --no-cpu-baseline > $O/bench_r04_attn_asm_pw64.json 2> $O/bench_r04_attn_asm.err
python - <<PY
import json
for f in ("bench_r04_c4_vitl14_b16", "bench_r04_c5_fp8", "bench_r04_c1_vits8_b2", "bench_r04_attn_asm_pw64"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("hbm_row", {}).get("frac"), d["roofline"].get("attention_row", {}).get("frac"), d["roofline"].get("attention_row", {}).get("avg_launch_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -12 $O/r04_per_step_calls.txt; head -20 $O/prof_r04_lora_bench.txt; cat $O/pmc_r04_traffic.json | head -30
