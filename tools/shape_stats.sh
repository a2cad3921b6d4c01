#!/bin/bash
# per-kernel times of one real-data shape: tools/shape_stats.sh genome|coverage [scale]
set -u
R=$(pwd); OUT=$R/gpurun_out/shape_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/shapes.py $1 --scale ${2:-0.2} --passes 2 > $OUT/run.log 2>&1
rm -f $OUT/*/*_kernel_trace.csv
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), "%8.2f ms total" % (float(r["TotalDurationNs"]) / 1e6), r["Percentage"])
PY
tail -1 $OUT/run.log | cut -c1-400
