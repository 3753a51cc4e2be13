#!/bin/bash
# rocprofv3 kernel-trace statistics of the assembly GEMM (form 0) and of the product kernel on the same shape (each run under its own timeout)
R=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/gemm_asm_prof
for form in 0 -1; do
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gemm_asm_prof/f$form -- python3 $R/tools/attn_asm/gpu_check_gemm.py one $form 30 43520x2304 > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/gemm_asm_prof/*/*/*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Name"]:
            print(f.split("/")[2], r["Name"][:70], "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
