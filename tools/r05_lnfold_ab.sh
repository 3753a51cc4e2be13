#!/bin/bash
# Round 5: LayerNorm folded into QKV / fc1 (fp16-operand build): tests, then interleaved bench runs with the fold on / off.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lnfold.py -x -q 2>&1 | tail -15 > gpurun_out/r05_lnfold_tests.txt
python -m pytest tests/test_gpu_parity_c2.py -q -k "c2_full_size or massive or batch" 2>&1 | tail -15 >> gpurun_out/r05_lnfold_tests.txt
for i in 1 2; do
  for f in on off; do
    python bench.py --half f16 --resid f16 --ln-fold $f --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_f16_fold_${f}_$i.json 2> gpurun_out/r05_f16_fold_${f}_$i.err
  done
done
python bench.py --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_bf16_ref.json 2> gpurun_out/r05_bf16_ref.err
grep -h -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"' gpurun_out/r05_f16_fold_*.json gpurun_out/r05_bf16_ref.json
cat gpurun_out/r05_lnfold_tests.txt
