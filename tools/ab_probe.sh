#!/bin/bash
# same-box A/B of an environment knob: tools/ab_probe.sh ENVVAR [rounds] [bench flags]; per run: ms per pass, dominant kernel's average
# launch (us), insert / simplitig / overlap phase (ms).  "on" = the variable set to 1.
V=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for mode in off on; do
    if [ $mode = on ]; then export $V=1; else unset $V; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 4 "$@" 2>&1 | tail -1 > /tmp/line.json
    MODE=$mode VAR=$V python3 - <<'PY'
import json, os
d = json.load(open('/tmp/line.json')); p = d["phases_ms"]
print(os.environ["VAR"], os.environ["MODE"], d["ms_per_step"], d["roofline"]["avg_launch_us"], p["us_insert_phase"], p["us_simplitig_phase"], p["us_overlap_phase"])
PY
  done
done
