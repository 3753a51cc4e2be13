#!/bin/bash
# Round 5: LayerNorm folded into QKV / fc1 (fp16-operand build): tests, then interleaved bench runs: fold with row partials / fold with the statistics kernel / no fold.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_lnfold.py -x -q 2>&1 | tail -15 > gpurun_out/r05_lnfold_tests.txt
python -m pytest tests/test_gpu_parity_c2.py -q -k "c2_full_size or massive or batch" 2>&1 | tail -15 >> gpurun_out/r05_lnfold_tests.txt
for i in 1 2; do
  python bench.py --half f16 --resid f16 --ln-fold on --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_f16_fold_part_$i.json 2> gpurun_out/r05_f16_fold_part_$i.err
  UCOD_LN_FOLD_NO_PARTIALS=1 python bench.py --half f16 --resid f16 --ln-fold on --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_f16_fold_stats_$i.json 2> gpurun_out/r05_f16_fold_stats_$i.err
  python bench.py --half f16 --resid f16 --ln-fold off --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_f16_fold_off_$i.json 2> gpurun_out/r05_f16_fold_off_$i.err
  python bench.py --no-cpu-baseline --lora-steps 0 --steps 40 > gpurun_out/r05_bf16_ref_$i.json 2> gpurun_out/r05_bf16_ref_$i.err
done
for f in gpurun_out/r05_f16_fold_part_?.json gpurun_out/r05_f16_fold_stats_?.json gpurun_out/r05_f16_fold_off_?.json gpurun_out/r05_bf16_ref_?.json; do echo -n "$f "; grep -h -o '"value": [0-9.]*, "unit": "images/s", "n_gpus"' $f; done
cat gpurun_out/r05_lnfold_tests.txt
