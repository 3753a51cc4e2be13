#!/bin/bash
# full GPU suite + bench line + the round's rocprof / PMC evidence at the current commit
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03h_pytest.txt 2>&1
tail -4 gpurun_out/r03h_pytest.txt
timeout 900 python bench.py > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.err
tail -c 600 gpurun_out/r03h_bench.json
timeout 1500 bash tools/r03_evidence.sh > gpurun_out/r03h_evidence.log 2>&1
tail -5 gpurun_out/r03h_evidence.log
ls gpurun_out | head -50
