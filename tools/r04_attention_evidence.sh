#!/bin/bash
# Round-4 attention evidence (GPU box, repo root): tests of the assembly variants, interleaved A/B on random and on constant data,
# cycle stamps, ablations, PMC passes.  Writes gpurun_out/r04_attn/*.txt (copied to profiles/ by hand).
mkdir -p gpurun_out/r04_attn
O=gpurun_out/r04_attn
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" -x 2>&1 | tail -3 > $O/tests.txt
ATTN_ROUNDS=9 ATTN_ITERS=20 timeout 300 python tools/attn_ab.py v5=product:5 pw64=product:64 pw32=product:32 2>&1 | grep -v amdgpu.ids > $O/ab.txt
ATTN_ROUNDS=7 ATTN_ITERS=10 timeout 400 python tools/attn_asm/bench_variants.py pw64= pw64_stamps=stamps pw32=pw32 pw32_stamps=pw32+stamps pw32_noskip=pw32+noskip pw64_msum=msum pw64_unitdetect=ud1 pw64_ring8=ring8 2>&1 | grep -v amdgpu.ids > $O/forms.txt
ATTN_ZERO=1 ATTN_ROUNDS=5 ATTN_ITERS=10 timeout 300 python tools/attn_asm/bench_variants.py pw64= pw32=pw32 2>&1 | grep -v amdgpu.ids > $O/zero_data.txt
timeout 600 python tools/attn_asm/bench_variants.py full= mfma_only=nosm+nolds+nodma+nobar valu_only=nomfma+nolds+nodma+nobar nomfma=nomfma nosm=nosm nolds=nolds nodma=nodma nobar=nobar exp2mov=exp2mov 2>&1 | grep -v amdgpu.ids > $O/ablation_pw64.txt
timeout 600 python tools/attn_asm/bench_variants.py full=pw32 mfma_only=pw32+nosm+nolds+nodma+nobar valu_only=pw32+nomfma+nolds+nodma+nobar nosm=pw32+nosm nolds=pw32+nolds nodma=pw32+nodma nobar=pw32+nobar 2>&1 | grep -v amdgpu.ids > $O/ablation_pw32.txt
timeout 600 bash tools/attn_pmc.sh 64 ucod_attn_fwd_pw64 > $O/pmc_pw64.txt 2>&1
timeout 600 bash tools/attn_pmc.sh 32 ucod_attn_fwd_pw32 > $O/pmc_pw32.txt 2>&1
timeout 600 bash tools/attn_pmc.sh 5 attn_fwd_v5 > $O/pmc_v5.txt 2>&1
tail -n 50 $O/tests.txt $O/ab.txt $O/forms.txt $O/zero_data.txt
