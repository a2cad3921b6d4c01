#!/bin/bash
# A/B of the block-build kernel with / without the id preparation on its way: per-pass phases of the default bench
for v in "KC_NO_BUILD_IDS=1" "KC_X=0" "KC_X=1"; do
  echo "== $v"
  env $v python3 bench.py --no-cpu-baseline --no-cli --steps 2 --warmup 1 > gpurun_out/ab.out 2> gpurun_out/ab.err
  echo "rc=$?"
  grep '^{"metric"' gpurun_out/ab.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'])"
done
