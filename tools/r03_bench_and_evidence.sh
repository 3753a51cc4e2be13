#!/bin/bash
# bench line + the round's rocprof / PMC evidence at the current commit (the GPU suite is run separately)
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r03_serial gpurun_out/prof_r03_default gpurun_out/prof_r03_lora gpurun_out/pmc_r03_fetch gpurun_out/pmc_r03_write
timeout 900 python bench.py > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err
tail -c 400 gpurun_out/r03i_bench.json
timeout 1500 bash tools/r03_evidence.sh > gpurun_out/r03i_evidence.log 2>&1
tail -3 gpurun_out/r03i_evidence.log
