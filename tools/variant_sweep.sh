#!/bin/bash
run() {
  python bench.py "$@" --lora-steps 0 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/b.json
  python - "$*" <<'PY'
import json, sys
try:
    d = json.load(open("/tmp/b.json"))
    print(f"{sys.argv[1]:36s} {d['value']:8.1f} img/s {d['ms_per_step']:7.3f} ms  serial {d['roofline']['serial_ms_per_step']} ms")
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run --gemm-variant 0
run --gemm-variant 5
run --gemm-variant 6
run --gemm-variant 7
run --gemm-variant 8
