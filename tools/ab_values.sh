#!/bin/bash
# same-box sweep of one environment knob over several values: tools/ab_values.sh ENVVAR "v1 v2 ..." [rounds] [bench flags]
V=$1; VALS=$2; N=${3:-2}; shift; shift; shift
for i in $(seq $N); do
  for val in $VALS; do
    if [ "$val" = "-" ]; then unset $V; else export $V=$val; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 4 "$@" 2>&1 | tail -1 > /tmp/line.json
    VAL=$val VAR=$V python3 - <<'PY'
import json, os
d = json.load(open('/tmp/line.json')); p = d["phases_ms"]
print(os.environ["VAR"], os.environ["VAL"], d["ms_per_step"], d["roofline"]["avg_launch_us"], p["us_insert_phase"], p["us_simplitig_phase"], p["us_overlap_phase"], p["us_emit_phase"])
PY
  done
done
