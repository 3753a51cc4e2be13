#!/bin/bash
# round-5 evidence on the GPU box (from the repo root).  Summaries land in gpurun_out/; the ones to keep are copied into profiles/ by hand.
#   * rocprofv3 kernel stats of the headline configuration (fp16 operands, fp16 stream, LayerNorm folded): serial and default schedule; PMC traffic
#   * the bf16 configuration's serial stats beside it
#   * launch order of one serial step (rocclr fills / copies), LoRA-mode stats
#   * side configurations: C4 first stage, C4 Look-Twice leg, C5 fp8, C1; next-rows bench
set -u
# (round 6, ADVICE r5: every rocprofv3 pass runs under `timeout` -- a FETCH_SIZE / WRITE_SIZE pass once hung a box for 25 minutes; a pass that times out is skipped)
R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/refresh_evidence.sh r05 > $O/r05_refresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
SERIAL="--steps 5 --warmup 2 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_bf16_serial -- python3 $R/bench.py --half bf16 $SERIAL > $O/prof_r05_bf16_serial_bench.json 2> $O/prof_r05_bf16_serial.err
find $O/prof_r05_bf16_serial -name '*kernel_trace.csv' -delete
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace_r05 -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 > /dev/null 2> $O/trace_r05.err
python3 $R/tools/step_trace.py $O/trace_r05 > $O/r05_step_trace.txt 2>&1
rm -rf $O/trace_r05
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r05_lora -- python3 $R/tools/lora_bench.py 32 2 1 0.05 > $O/prof_r05_lora_bench.txt 2> $O/prof_r05_lora.err
find $O/prof_r05_lora -name '*kernel_trace.csv' -delete
cd $R
python bench.py --arch dinov2_vitl14 --batch 16 --lora-steps 0 --cpu-images 1 > $O/bench_r05_c4_vitl14_b16.json 2> $O/bench_r05_c4.err
python bench.py --look-twice --arch dinov2_vitl14 --batch 16 --steps 10 --warmup 2 > $O/bench_r05_c4_look_twice.json 2> $O/bench_r05_c4_lt.err
python bench.py --look-twice --arch dinov2_vitl14 --batch 16 --steps 10 --warmup 2 --half bf16 > $O/bench_r05_c4_look_twice_bf16.json 2> $O/bench_r05_c4_lt_bf16.err
python bench.py --half bf16 --batch 64 --attn-variant 8 --lora-steps 0 --cpu-images 1 > $O/bench_r05_c5_fp8.json 2> $O/bench_r05_c5.err
python bench.py --arch dino_vits8 --image 224 --batch 2 --lora-steps 0 --cpu-images 1 > $O/bench_r05_c1_vits8_b2.json 2> $O/bench_r05_c1.err
python tools/next_rows_bench.py > $O/r05_next_rows_bench.txt 2> $O/r05_next_rows.err
python - <<PY
import json
for f in ("bench_r05_c4_vitl14_b16", "bench_r05_c4_look_twice", "bench_r05_c4_look_twice_bf16", "bench_r05_c5_fp8", "bench_r05_c1_vits8_b2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("hbm_row", {}).get("frac"), d["roofline"].get("attention_row", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $O/r05_step_trace.txt | head -20; head -12 $O/prof_r05_lora_bench.txt; head -40 $O/pmc_r05_traffic.json; tail -20 $O/r05_next_rows_bench.txt
