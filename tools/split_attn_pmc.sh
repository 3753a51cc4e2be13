#!/bin/bash
# Matrix-pipe occupancy, instruction mix and LDS conflicts of the split attention kernel (two / three terms) at C2: rocprofv3 --pmc passes (each under timeout) over
# tools/split_bench.py.  Run on the GPU box from the repo root: bash tools/split_attn_pmc.sh  ->  gpurun_out/r06_split_attn_pmc.txt
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/split_attn_pmc_$i -- python3 $R/tools/split_bench.py --terms 2 --iters 3 > /dev/null 2> $O/split_attn_pmc_$i.err || echo "pass $i skipped"
done
cd $R; python3 - <<'PY' > gpurun_out/r06_split_attn_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/split_attn_pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for name in ("attn_split_kernel", "qkv_split_kernel", "layernorm_split_kernel"):
            if name in r["Kernel_Name"]:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc over tools/split_bench.py --terms 2 (means per launch; 32 x 12 x 1370 tokens)")
for name, c in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print(name, {k: round(v, 1) for k, v in sorted(m.items())})
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        print("   MFMA busy = %.3f of the chip's matrix-pipe cycles while the kernel runs" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] / 32 * 1024)))
    if "SQ_INSTS_VALU" in m and m.get("SQ_INSTS_MFMA", 0) > 0:
        print("   vector instructions per MFMA = %.2f ; LDS instructions per MFMA = %.2f" % ((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"], m.get("SQ_INSTS_LDS", 0) / m["SQ_INSTS_MFMA"]))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)))
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
        print("   wait_any / wave cycles = %.3f" % (m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
rm -rf $O/split_attn_pmc_[0-9]
cat gpurun_out/r06_split_attn_pmc.txt
