#!/bin/bash
# phase-2 time against the switch-over point of the grouped levels (KC_OVG_MAX), cfg2 and the coverage shape
for m in 0 262144 1048576 2097152 4194304; do
  echo "== KC_OVG_MAX=$m"
  KC_OVG_MAX=$m python3 bench.py --no-cpu-baseline --no-cli --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['ms_per_step'], d['phases_ms']['us_overlap_phase'])"
  KC_OVG_MAX=$m python3 tools/time_coverage_reads.py 2>&1 | tail -1 | sed 's/.*device ms/device ms/'
done
