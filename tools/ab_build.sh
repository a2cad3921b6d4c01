#!/bin/bash
# A/B of the block-build kernel's write-out (16-byte stores, two slots per lane / 8-byte stores): insert phase of the default bench, alternating
for v in "KC_BUILD_ST8=1" "KC_X=0" "KC_BUILD_ST8=1" "KC_X=0"; do
  echo "== $v"
  env $v timeout 300 python3 bench.py --no-cpu-baseline --no-cli --steps 3 --warmup 1 > gpurun_out/ab.out 2> gpurun_out/ab.err
  echo "rc=$?"
  grep '^{"metric"' gpurun_out/ab.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'])"
done
