#!/bin/bash
# per-kernel times (top of the rocprofv3 summary) of one benchmark configuration: tools/kernel_stats.sh [bench flags]
set -u
R=$(pwd); OUT=$R/gpurun_out/kernel_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli "$@" > $OUT/bench.log 2>&1
rm -f $OUT/*/*_kernel_trace.csv
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
