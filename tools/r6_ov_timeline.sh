#!/bin/bash
# phase 2 of one pass by kernel (tools/ov_timeline.py over a kernel trace): tools/r6_ov_timeline.sh [bench flags]
set -u
R=$(pwd); OUT=$R/gpurun_out/r6_ov_timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli "$@" > $OUT/bench.log 2>&1
python3 $R/tools/ov_timeline.py $OUT
rm -f $OUT/*/*_kernel_trace.csv
