#!/bin/bash
# which runtime copy kernels (__amd_rocclr_copyBuffer) does one pass issue, how long do they take, and after which kernel?
set -u
R=$(pwd); OUT=$R/gpurun_out/trace_copies; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
rows = rows[half:]                     # roughly the timed pass
prev = None; agg = {}
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "copyBuffer" in n or "fillBuffer" in n:
        key = (n[:28], (prev or "")[:40])
        a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += d; a[2] = max(a[2], d)
    prev = n
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-28s after %-40s calls %4d total %9.1f us max %8.1f us" % (k[0], k[1], a[0], a[1], a[2]))
PY
rm -f $OUT/*/*_kernel_trace.csv
