#!/bin/bash
# bench.py with 1..4 image-parallel backbone streams
for s in 1 2 3 4; do
  python bench.py --streams $s --lora-steps 0 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > /tmp/b_$s.json
  python - "$s" <<'PY'
import json, sys
d = json.load(open(f"/tmp/b_{sys.argv[1]}.json"))
print("streams", sys.argv[1], d["value"], "img/s", d["ms_per_step"], "ms", "loss", d["final_loss"])
PY
done
