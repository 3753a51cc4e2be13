#!/bin/bash
# Run on the MI355X box from the repo root: regenerates the rocprofv3 summaries and PMC traffic that profiles/ keeps per state.
# usage: bash tools/refresh_evidence.sh <state label, e.g. r01x>
set -u
# (round 6, ADVICE r5: every rocprofv3 pass runs under `timeout` -- a FETCH_SIZE / WRITE_SIZE pass once hung a box for 25 minutes; a pass that times out is skipped)
S=${1:-r01x}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SERIAL="--steps 5 --warmup 2 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 --sustain-s 0"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${S}_serial -- python3 $R/bench.py $SERIAL > $O/prof_${S}_serial_bench.json 2> $O/prof_${S}_serial.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${S}_default -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --lora-steps -1 --sustain-s 0 > $O/prof_${S}_default_bench.json 2> $O/prof_${S}_default.err
PM="--steps 2 --warmup 1 --no-cpu-baseline --lora-steps -1 --no-pipeline --streams 1 --sustain-s 0"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${S}_fetch -- python3 $R/bench.py $PM > /dev/null 2> $O/pmc_${S}_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${S}_write -- python3 $R/bench.py $PM > /dev/null 2> $O/pmc_${S}_write.err
python3 $R/tools/pmc_traffic.py $O/pmc_${S}_fetch $O/pmc_${S}_write $S > $O/pmc_${S}_traffic.json
# keep only the summaries (the raw counter CSVs are tens of MB)
find $O/pmc_${S}_fetch $O/pmc_${S}_write -name '*counter_collection.csv' -delete
find $O/prof_${S}_serial $O/prof_${S}_default -name '*kernel_trace.csv' -delete
ls -la $O/prof_${S}_serial/*/ $O/pmc_${S}_traffic.json
