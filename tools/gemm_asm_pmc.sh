#!/bin/bash
# (every pass under its own `timeout`: a FETCH_SIZE / WRITE_SIZE pass over this script hung a box for 25 minutes in round 5 and is not in the list)
# PMC passes over ONE form of the assembly GEMM (or the product kernel: form -1) at the QKV shape.  Run on the GPU box from the repo root:
#   bash tools/gemm_asm_pmc.sh "<forms>" [MxN] > profiles/rNN_gemm_asm_pmc.txt         (separate --pmc passes, kernel trace only)
FORMS=${1:-"0 -1"}; SHAPE=${2:-43520x2304}
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/gemm_asm_pmc
for form in $FORMS; do
  i=0
  for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" "TA_BUSY_avr TD_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/gemm_asm_pmc/f${form}_$i -- python3 $R/tools/attn_asm/gpu_check_gemm.py one $form 4 $SHAPE > /dev/null 2>&1
  done
done
cd $R; python3 - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/gemm_asm_pmc/*/*/*counter_collection.csv"):
    form = re.search(r"gemm_asm_pmc/f(-?\d+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_pk" in k or "gemm_bf16_mixed_kernel" in k or "gemm_bf16_big_kernel" in k:
            agg[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/gemm_asm_pmc/*/*/*kernel_trace.csv"):
    form = re.search(r"gemm_asm_pmc/f(-?\d+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_pk" in k or "gemm_bf16_mixed_kernel" in k or "gemm_bf16_big_kernel" in k:
            dur[form].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for form in sorted(agg, key=int):
    m = {k: sum(v) / len(v) for k, v in agg[form].items()}
    us = sum(dur[form]) / max(1, len(dur[form]))
    print(f"== form {form} (-1 = the product kernel): {us:.1f} us under the profiler")
    for k in sorted(m):
        print(f"   {k:32s} {m[k]:.6g}")
    if "SQ_BUSY_CYCLES" in m:
        cyc = m["SQ_BUSY_CYCLES"] / 32.0
        print(f"   cycles {cyc:.4g}, matrix pipe busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):.3f}")
    if "TCC_HIT_sum" in m:
        print(f"   L2 hit rate {m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
PY
