#!/bin/bash
# Collects, on the GPU box, the evidence behind bench.py's roofline numbers (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command        -> <out>/stats
#   2. two separate PMC passes (FETCH_SIZE ; WRITE_SIZE + L2 hit/miss), as MI355X_MICROARCH.md (HBM, rocprofv3 PMC slots)
#      prescribes: FETCH_SIZE needs 3 of the 4 TCC slots, so it cannot share a pass with WRITE_SIZE.
# The program after `--` is python3 itself (no env/bash hops under the profiler).
set -u
OUT=${1:-gpurun_out/profile}
shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-cli}
R=$(pwd)
# the counter passes run one step of the same workload (the workload flags are taken from ARGS, steps / warmup overridden)
PMCARGS="$ARGS --steps 1 --warmup 0"
rm -rf $R/$OUT; mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py $ARGS > $R/$OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -- python3 $R/bench.py $PMCARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/$OUT/pmc_write -- python3 $R/bench.py $PMCARGS > /dev/null 2>&1
rm -f $R/$OUT/*/*/*_kernel_trace.csv
tail -1 $R/$OUT/bench_stats.log
