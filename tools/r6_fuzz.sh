#!/bin/bash
# round 6: the randomised parity run on the final build -- several processes side by side on the one GPU (the cases are small: the box's
# cores and launch latencies bound a single process), then the CLI's `--gpus N` fuzz.  Usage (gpurun, repo root): bash tools/r6_fuzz.sh [seconds] [processes]
SECS=${1:-420}; NP=${2:-6}
mkdir -p gpurun_out/r6_fuzz
for i in $(seq $NP); do
  timeout $((SECS + 120)) python3 tools/fuzz_parity.py $SECS $((${3:-6100} + i)) > gpurun_out/r6_fuzz/fuzz_$i.log 2>&1 &
done
wait
for i in $(seq $NP); do tail -2 gpurun_out/r6_fuzz/fuzz_$i.log | cut -c1-300; done
timeout 300 python3 tools/fuzz_cli_gpus.py 180 81 > gpurun_out/r6_fuzz/fuzz_gpus.log 2>&1
echo "fuzz gpus rc=$?"; tail -2 gpurun_out/r6_fuzz/fuzz_gpus.log | cut -c1-300
