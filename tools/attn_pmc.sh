#!/bin/bash
# PMC passes over the attention micro-benchmark.  Run on the GPU box from the repo root:
#   bash tools/attn_pmc.sh [variant] [kernel-name substring] > profiles/rNN_attention_pmc.txt
# defaults: variant 2 = attn_fwd_v5_kernel, the product path; 9 + attn_fwd_pp = the 8-wave ping-pong form.  Separate --pmc passes.
V=${1:-2}; KN=${2:-attn_fwd_v5}; export KN
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/attn_pmc
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc/$tag -- python3 $R/tools/attn_bench.py $V 3 > /dev/null 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, collections, os
KN = os.environ['KN']
agg = collections.defaultdict(list)
dur = []
for f in glob.glob("gpurun_out/attn_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if KN in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/attn_pmc/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if KN in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(f"# attention forward ({KN} kernel) at C2: B=32, 12 heads, N=1370, head_dim 64; means per launch over the profiled launches")
for k in sorted(m):
    print(f"{k:32s} n={len(agg[k]):3d} mean={m[k]:.6g}")
us = sum(dur) / max(1, len(dur))
print(f"kernel duration under the profiler: {us:.1f} us  ({184.5e9 / (us * 1e-6) / 1e12:.0f} TFLOP/s)")
# units (MI355X_MICROARCH.md, cycle-constants table): SQ_BUSY_CYCLES is summed over 32 shader engines; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES /
# SQ_WAIT_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.
if "SQ_BUSY_CYCLES" in m:
    cyc = m["SQ_BUSY_CYCLES"] / 32.0
    simd_cycles = cyc * 1024
    print(f"kernel cycles (SQ_BUSY_CYCLES / 32)          : {cyc:.4g}  -> clock {cyc / us / 1e3:.2f} GHz")
    print(f"matrix pipe busy  (MFMA_BUSY / SIMD-cycles)  : {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}")
    print(f"VALU issue busy   (4*ACTIVE_INST_VALU / SIMD-cycles): {4 * m['SQ_ACTIVE_INST_VALU'] / simd_cycles:.3f}   (includes the MFMAs' own issue cycles)")
    if "SQ_INSTS_VALU_TRANS_F32" in m:
        print(f"transcendental issue (8 cyc each / SIMD-cycles): {8 * m['SQ_INSTS_VALU_TRANS_F32'] / simd_cycles:.3f}")
    if "SQ_INSTS_VALU" in m:
        print(f"VALU instructions per MFMA                   : {m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']:.2f}  (exp per MFMA {m.get('SQ_INSTS_VALU_TRANS_F32', 0) / m['SQ_INSTS_MFMA']:.2f})")
    if "SQ_WAVE_CYCLES" in m:
        print(f"wave occupancy (4*WAVE_CYCLES / SIMD-cycles)  : {4 * m['SQ_WAVE_CYCLES'] / simd_cycles:.2f} waves per SIMD")
PY
