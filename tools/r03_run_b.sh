#!/bin/bash
mkdir -p gpurun_out
python tools/attn_ab.py r2=lab:102 f1=ucod_dpl_amd/_native/libucod_dpl_forms.so:21 f2=ucod_dpl_amd/_native/libucod_dpl_forms.so:22 f3=ucod_dpl_amd/_native/libucod_dpl_forms.so:23 f4=ucod_dpl_amd/_native/libucod_dpl_forms.so:24 product=product:2 > gpurun_out/r03b_attn_ab.txt 2>&1
tail -8 gpurun_out/r03b_attn_ab.txt
rm -f gpurun_out/parity_c2_measured.jsonl
timeout 2000 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/r03b_pytest.txt 2>&1
tail -30 gpurun_out/r03b_pytest.txt
