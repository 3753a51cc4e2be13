#!/bin/bash
# (round 6, ADVICE r5: every rocprofv3 pass runs under `timeout`; a pass that times out is skipped)
# Matrix-pipe occupancy and instruction mix of the step's kernels (headline configuration and the bf16 one), from rocprofv3 --pmc passes over the serial step.
# Run on the GPU box from the repo root: bash tools/r05_step_pmc.sh  ->  gpurun_out/r05_step_pmc.txt
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for half in f16 bf16; do
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/step_pmc_${half}_$i -- python3 $R/bench.py --half $half --steps 2 --warmup 1 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 --sustain-s 0 > /dev/null 2> $O/step_pmc_${half}_$i.err
  done
done
cd $R; python3 - <<'PY' > gpurun_out/r05_step_pmc.txt
import csv, glob, collections
CLS = [("qkv", ("mixed_kernel<0,", "mixed_kernel<11,")), ("fc1", ("mixed_kernel<1,", "mixed_kernel<12,")), ("out-proj + fc2", ("mixed_kernel<9,", "mixed_kernel<13,")),
       ("attention", ("attn_fwd_v5_kernel",)), ("layernorm", ("layernorm_h16_strip",)), ("row statistics", ("row_stats_h16",)), ("key hook", ("big_kernel<4,",)), ("patch embedding", ("big_kernel<10,", "big_kernel<14,"))]
print("# rocprofv3 --pmc over `bench.py --half <h> --steps 2 --warmup 1 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1` (tools/r05_step_pmc.sh), means per launch")
print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024): fraction of the chip's matrix-pipe cycles in use while the kernel runs")
for half in ("f16", "bf16"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/step_pmc_{half}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for name, subs in CLS:
                if any(s in r["Kernel_Name"] for s in subs):
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"## {half} operands" + (" (LayerNorm folded into QKV / fc1)" if half == "f16" else ""))
    for name, _ in CLS:
        c = {n: sum(v) / len(v) for n, v in agg[name].items()}
        if "SQ_BUSY_CYCLES" not in c:
            continue
        cyc = c["SQ_BUSY_CYCLES"] / 32
        n = len(agg[name]["SQ_BUSY_CYCLES"])
        print(f"{name:18s} launches {n:4d}  cycles/launch {cyc:9.0f}  MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024):5.2f}  MFMA insts {c.get('SQ_INSTS_MFMA', 0):.3g}  "
              f"VALU insts {c.get('SQ_INSTS_VALU', 0):.3g}  VALU per MFMA {c.get('SQ_INSTS_VALU', 0) / max(c.get('SQ_INSTS_MFMA', 0), 1):6.2f}  LDS insts {c.get('SQ_INSTS_LDS', 0):.3g}  "
              f"wait_any / wave cycles {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):5.2f}")
PY
find $O -path '*step_pmc_*' -name '*counter_collection.csv' -delete
cat gpurun_out/r05_step_pmc.txt
