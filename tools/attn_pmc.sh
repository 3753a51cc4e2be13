#!/bin/bash
# PMC passes over the attention micro-benchmark (variant 2).  Run on the GPU box: bash tools/attn_pmc.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/attn_pmc/$tag -- python3 $R/tools/attn_bench.py 2 3 > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/attn_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_fwd_v2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
