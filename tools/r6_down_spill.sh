#!/bin/bash
# round 6 timing experiment (tools' build, results stay right): the probe sweep without the cross-block lookups towards EARLIER blocks
# (what an "only one end of a cross-block edge looks it up" scheme would leave in the sweep).  Through gpurun, repo root.
export KMERCAMEL_AMD_LIB=$(pwd)/kmercamel_amd/libkmercamel_amd_exp.so
for i in 1 2; do
  for knob in none KC_PROBE_DOWN_SPILL; do
    if [ $knob != none ]; then export $knob=1; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 2 --warmup 1 2>&1 | tail -1 > /tmp/line.json
    if [ $knob != none ]; then unset $knob; fi
    MODE=$knob python3 - <<'PY'
import json, os
d = json.load(open('/tmp/line.json')); p = d["phases_ms"]
print(os.environ["MODE"], d["ms_per_step"], d["roofline"]["avg_launch_us"], p["us_insert_phase"], p["us_simplitig_phase"], p["us_overlap_phase"])
PY
  done
done
