#!/bin/bash
# same-box A/B of one environment knob over several values: tools/ab_values.sh VAR v1 v2 ...  (overlap phase and pass of the default bench)
var=$1; shift
for v in "$@" "$1"; do
  echo "== $var=$v"
  env $var=$v timeout 300 python3 bench.py --no-cpu-baseline --no-cli --steps 3 --warmup 1 > gpurun_out/ab.out 2> gpurun_out/ab.err
  grep '^{"metric"' gpurun_out/ab.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['us_overlap_phase'])"
done
