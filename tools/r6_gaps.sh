#!/bin/bash
# idle gaps of one pass (tools/r6_gaps.py over a kernel trace): tools/r6_gaps.sh [bench flags]
set -u
R=$(pwd); OUT=$R/gpurun_out/r6_gaps; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli "$@" > "$OUT/bench.log" 2>&1
cd $R
python3 tools/r6_gaps.py "$OUT"
find "$OUT" -name '*_kernel_trace.csv' -delete
