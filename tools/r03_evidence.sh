#!/bin/bash
# round-3 evidence on the GPU box (from the repo root): rocprofv3 kernel stats (serial, default, LoRA mode), PMC traffic, attention PMC,
# bf16-vs-fp16 clock pair.  Summaries land in gpurun_out/; the ones to keep are copied into profiles/ by hand.
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
bash tools/refresh_evidence.sh r03 > $O/r03_refresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
# backbone-backward mode, one stream (exclusive kernel times)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_lora -- python3 $R/tools/lora_bench.py 32 2 1 > $O/prof_r03_lora_bench.txt 2> $O/prof_r03_lora.err
find $O/prof_r03_lora -name '*kernel_trace.csv' -delete
# bf16 vs fp16 operands: duration and clock per kernel class
PM="--steps 2 --warmup 1 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 --resid f16"
for h in bf16 f16; do
  rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/clock_r03_$h -- python3 $R/bench.py --half $h $PM > /dev/null 2> $O/clock_r03_$h.err
done
python3 $R/tools/clock_pair.py $O/clock_r03_bf16 $O/clock_r03_f16 > $O/r03_f16_vs_bf16_clock.txt
find $O/clock_r03_bf16 $O/clock_r03_f16 -name '*.csv' -delete
cd $R
bash tools/attn_pmc.sh 2 attn_fwd_v5 > $O/r03_attention_pmc.txt 2> $O/r03_attention_pmc.err
rm -rf $O/attn_pmc
cat $O/r03_f16_vs_bf16_clock.txt; tail -25 $O/r03_attention_pmc.txt; cat $O/pmc_r03_traffic.json | head -40; cat $O/prof_r03_lora_bench.txt | head -30
