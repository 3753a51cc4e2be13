#!/bin/bash
# round-6 evidence on the GPU box (from the repo root).  Summaries land in gpurun_out/; the ones to keep are copied into profiles/ by hand.
# Every rocprofv3 pass runs under `timeout` (a counter pass once hung a box for 25 minutes); a pass that times out is skipped and says so.
#   1. the driver-shaped line: python bench.py (40-s sustained run, five configurations in their own processes)        -> bench_r06.json
#   2. the same with the roofline's HBM traffic measured by the run (--pmc-traffic)                                      -> bench_r06_pmc.json
#   3. rocprofv3 --kernel-trace --stats of the serial step (the pair the roofline is checked against) and of the default schedule; PMC traffic file
#   4. the split-operand configurations' serial stats
#   5. side configurations: C4 first stage, C4 Look-Twice leg, C5 fp8, C1; UCOD_FORCE_DIST=1 line
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
STAGE=${1:-all}
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  python bench.py > $O/bench_r06.json 2> $O/bench_r06.err
  UCOD_FORCE_DIST=1 python bench.py --no-cpu-baseline --lora-steps -1 --sustain-s 20 > $O/bench_r06_force_dist.json 2> $O/bench_r06_force_dist.err
fi
if [ "$STAGE" = all ] || [ "$STAGE" = pmc ]; then
  python bench.py --pmc-traffic --lora-steps -1 --no-cpu-baseline --sustain-s 0 > $O/bench_r06_pmc.json 2> $O/bench_r06_pmc.err
fi
if [ "$STAGE" = all ] || [ "$STAGE" = prof ]; then
  bash tools/refresh_evidence.sh r06 > $O/r06_refresh.log 2>&1
  cd /tmp && export TMPDIR=/tmp
  SERIAL="--steps 5 --warmup 2 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 --sustain-s 0"
  for h in split2 split3; do
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_${h}_serial -- python3 $R/bench.py --half $h $SERIAL > $O/prof_r06_${h}_serial_bench.json 2> $O/prof_r06_${h}_serial.err || echo "$h pass skipped (timeout / error)"
    find $O/prof_r06_${h}_serial -name '*kernel_trace.csv' -delete
  done
  cd $R
fi
if [ "$STAGE" = all ] || [ "$STAGE" = side ]; then
  python bench.py --arch dinov2_vitl14 --batch 16 --lora-steps 0 --cpu-images 1 --sustain-s 0 > $O/bench_r06_c4_vitl14_b16.json 2> $O/bench_r06_c4.err
  python bench.py --look-twice --arch dinov2_vitl14 --batch 16 --steps 10 --warmup 2 > $O/bench_r06_c4_look_twice.json 2> $O/bench_r06_c4_lt.err
  python bench.py --half bf16 --batch 64 --attn-variant 8 --lora-steps 0 --cpu-images 1 --sustain-s 0 > $O/bench_r06_c5_fp8.json 2> $O/bench_r06_c5.err
  python bench.py --arch dino_vits8 --image 224 --batch 2 --lora-steps 0 --cpu-images 1 --sustain-s 0 > $O/bench_r06_c1_vits8_b2.json 2> $O/bench_r06_c1.err
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_r06*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["value"], d["ms_per_step"], d["dtype"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_at_held_clock"), (d.get("sustained") or {}).get("value_sustained"),
              {k: v["value"] for k, v in (d.get("configurations") or {}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
