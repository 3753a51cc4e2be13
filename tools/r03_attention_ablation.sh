#!/bin/bash
# attention-forward ablation table (lab variants 40-47, timing only) + the GPU suite + the bench line
mkdir -p gpurun_out
ATTN_ROUNDS=5 timeout 600 python tools/attn_ab.py product=product:2 plain=lab:30 no_mfma=lab:40 no_softmax=lab:41 no_ldsread=lab:42 no_dma=lab:43 \
  mfma_only=lab:44 valu_only=lab:45 mfma_valu=lab:46 mfma_mem=lab:47 > gpurun_out/r03g_attention_ablation.txt 2>&1
cat gpurun_out/r03g_attention_ablation.txt
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03g_pytest.txt 2>&1
tail -4 gpurun_out/r03g_pytest.txt
timeout 900 python bench.py > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
tail -c 1500 gpurun_out/r03g_bench.json
