#!/bin/bash
# round 6: the probe kernel's variants back to back on one box (boxes differ by ~7 %): parity subset first, then cfg[2] per knob.
# Usage (through gpurun, repo root): bash tools/r6_probe_ab.sh [rounds] [knobs...]     (default knobs: none KC_PROBE_NO_QUAD KC_PROBE_NO_SHARE KC_PROBE_NO_GUESS)
N=${1:-2}; shift
KNOBS=${@:-none KC_PROBE_NO_QUAD KC_PROBE_NO_SHARE KC_PROBE_NO_GUESS}
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu -k "simplitigs_match_canonical_oracle or alternative_device_paths or fragment_contraction or spneumoniae or segment_parallel or stress_reads or against_the_oracle" 2>&1 | tail -5
for i in $(seq $N); do
  for knob in $KNOBS; do
    if [ $knob != none ]; then export $knob=1; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 4 2>&1 | tail -1 > /tmp/line.json
    if [ $knob != none ]; then unset $knob; fi
    MODE=$knob python3 - <<'PY'
import json, os
d = json.load(open('/tmp/line.json')); p = d["phases_ms"]
print(os.environ["MODE"], d["ms_per_step"], d["roofline"]["avg_launch_us"], p["us_insert_phase"], p["us_simplitig_phase"], p["us_overlap_phase"])
PY
  done
done
