#!/bin/bash
# PMC passes over the GEMM micro-benchmark (auto variant).  Run on the GPU box: bash tools/gemm_pmc.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/gemm_pmc/$tag -- python3 $R/tools/gemm_bench.py 0 > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/gemm_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_bf16_big_kernel" in k or "gemm_bf16_mixed_kernel" in k:
            key = k.split("gemm_bf16_")[1][:28] + " grid" + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[key].items()}
    if "SQ_BUSY_CYCLES" not in c: continue
    cyc = c["SQ_BUSY_CYCLES"] / 32
    print(f"{key:34s} cycles/launch {cyc:9.0f}  MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):5.2f}  wait_any/wave {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.2f}"
          f"  wait_lds/wave {c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.2f}  lds_conflict/idx_active {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.3f}  mfma {c.get('SQ_INSTS_MFMA', 0):.3g}")
PY
