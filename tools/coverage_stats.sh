#!/bin/bash
# per-kernel times of the coverage-reads shape (tools/time_coverage_reads.py): tools/coverage_stats.sh [genome_Mbp] [coverage] [error_rate]
R=$(pwd); OUT=$R/gpurun_out/coverage_stats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KC_SIMP_TRACE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/time_coverage_reads.py ${1:-50} ${2:-10} ${3:-0.01} > $OUT/run.log 2>&1
grep -E 'reads x|\[simp\]' $OUT/run.log | tail -6 | cut -c1-250
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%10.1f us avg  total %9.1f ms" % (float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -f $OUT/*/*_kernel_trace.csv
