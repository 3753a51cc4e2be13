#!/bin/bash
# backbone-backward mode: student / teacher stream counts (bench.py's LoRA leg), two passes over the grid on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for cfg in "2 1" "1 1" "2 2" "3 1" "1 2"; do
    set -- $cfg
    UCOD_STUDENT_STREAMS=$1 UCOD_TEACHER_STREAMS=$2 timeout 300 python bench.py --no-cpu-baseline --lora-steps 8 --steps 4 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
b=d['backbone_backward_mode']
print('student $1 teacher $2:', b['value'], 'images/s', b['ms_per_step'], 'ms  (serial', b['serial_ms_per_step'], ')')"
  done
done
