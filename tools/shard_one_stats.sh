#!/bin/bash
# per-kernel times of one rank of the sharded flow (self transport) next to the single-GPU flow: tools/shard_one_stats.sh [records]
R=$(pwd); OUT=$R/gpurun_out/shard_one; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/shard_one.py ${1:-1000000} 2 > $OUT/run.log 2>&1
grep -E '^(sharded|single)' $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%10.1f us avg  total %9.1f ms" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -f $OUT/*/*_kernel_trace.csv
