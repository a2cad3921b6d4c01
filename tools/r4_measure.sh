#!/bin/bash
# short per-configuration lines: cfg2, cfg1, k63, k127 (ms per pass + phases) and the coverage-reads shape
for a in "" "--records 100000" "--records 100000 -k 63" "--records 100000 -k 127"; do
  python3 bench.py --no-cpu-baseline --no-cli --steps 3 --warmup 1 $a > gpurun_out/m.out 2> gpurun_out/m.err
  grep '^{"metric"' gpurun_out/m.out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'], d['phases_ms'], d['roofline']['avg_launch_us'])" || tail -3 gpurun_out/m.err
done
python3 tools/time_coverage_reads.py 2>&1 | tail -2
