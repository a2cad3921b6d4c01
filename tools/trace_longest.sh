#!/bin/bash
# the longest single launches of one pass (which kernel, how long) -- tools/trace_longest.sh [bench flags]
set -u
R=$(pwd); OUT=$R/gpurun_out/trace_longest; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
agg = {}
for r in rows:
    n = r["Kernel_Name"][:44]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(n, []).append(d)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
    v2 = sorted(v, reverse=True)
    print("%-44s n=%4d total %9.1f us  longest: %s" % (n, len(v), sum(v), " ".join("%.0f" % x for x in v2[:6])))
PY
rm -f $OUT/*/*_kernel_trace.csv
