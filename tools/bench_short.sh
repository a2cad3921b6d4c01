#!/bin/bash
# two short bench lines (cfg2, cfg1) reduced to: ms per pass, phases, average launch of the dominant kernel
for a in "" "--records 100000"; do
  python3 bench.py --no-cpu-baseline --no-cli --steps 3 $a 2>&1 | tail -1 > /tmp/line.json
  python3 - <<'PY'
import json
try:
    d = json.load(open('/tmp/line.json'))
    print(d["ms_per_step"], d["phases_ms"], d["roofline"]["avg_launch_us"])
except Exception as e:
    print("bench failed:", e, open('/tmp/line.json').read()[-2000:])
PY
done
