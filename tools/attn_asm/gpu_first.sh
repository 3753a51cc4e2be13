#!/bin/bash
# first contact: every stage in its own process under a timeout (a hang must not take the box)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for st in small branches lse; do
  timeout 180 python tools/attn_asm/gpu_check.py $st 2>&1 | tail -12
  echo "stage $st rc=$?"
done
ATTN_ROUNDS=5 ATTN_ITERS=10 timeout 300 python tools/attn_ab.py v5=product:5 asm64=product:64 asm32=product:32 2>&1 | tail -5
