#!/bin/bash
# same-box A/B of a probe-kernel knob: tools/ab_probe.sh ENVVAR [rounds]; prints the dominant kernel's average launch (us) per run
V=$1; N=${2:-3}
for i in $(seq $N); do
  for mode in off on; do
    if [ $mode = on ]; then export $V=1; else unset $V; fi
    python3 bench.py --no-cpu-baseline --no-cli --steps 4 2>&1 | tail -1 > /tmp/line.json
    python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('$V', '$mode', d['ms_per_step'], d['roofline']['avg_launch_us'])"
  done
done
