#!/bin/bash
mkdir -p gpurun_out
F=ucod_dpl_amd/_native/libucod_dpl_forms.so
python tools/attn_ab.py base=$F:30 seq5=$F:34 seq4=$F:35 product=product:2 > gpurun_out/r03e_attn_ab2.txt 2>&1
cat gpurun_out/r03e_attn_ab2.txt
