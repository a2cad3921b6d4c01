#!/bin/bash
# round 6: the probe kernel's variants back to back on one box (boxes differ by ~7 %): parity subset first, then cfg[2] with and without
# the in-block guess.  Usage (through gpurun, repo root): bash tools/r6_probe_ab.sh
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "simplitigs_match_canonical_oracle or alternative_device_paths or fragment_contraction or spneumoniae or segment_parallel or stress_reads" 2>&1 | tail -5
for i in 1 2; do
  for mode in guess noguess; do
    if [ $mode = noguess ]; then export KC_PROBE_NO_GUESS=1; else unset KC_PROBE_NO_GUESS; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 4 "$@" 2>&1 | tail -1 > /tmp/line.json
    MODE=$mode python3 - <<'PY'
import json, os
d = json.load(open('/tmp/line.json')); p = d["phases_ms"]
print(os.environ["MODE"], d["ms_per_step"], d["roofline"]["avg_launch_us"], p["us_insert_phase"], p["us_simplitig_phase"], p["us_overlap_phase"])
PY
  done
done
