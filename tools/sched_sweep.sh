#!/bin/bash
# bench.py under different schedules
run() {
  python bench.py "$@" --lora-steps 0 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/b.json
  python - "$*" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print(f"{sys.argv[1]:36s} {d['value']:8.1f} img/s {d['ms_per_step']:7.3f} ms  serial {d['roofline']['serial_ms_per_step']} ms  loss {d['final_loss']}")
PY
}
run --no-pipeline --streams 1
run --no-pipeline --streams 2
run --streams 1
run --streams 2
run --streams 3
