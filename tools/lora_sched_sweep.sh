#!/bin/bash
for cfg in "1 1" "2 1" "1 2" "2 2"; do
  set -- $cfg
  UCOD_STUDENT_STREAMS=$1 UCOD_TEACHER_STREAMS=$2 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > /tmp/b.json
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))["backbone_backward_mode"]
print("student/teacher streams", sys.argv[1], d["value"], "img/s", d["ms_per_step"], "ms")
PY
done
