#!/bin/bash
# kernel stats of `compute -z 2` on reads with coverage (tools/r5_z_coverage.py): tools/r5_z_prof.sh [genome_Mbp] [coverage]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5/zprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/tools/r5_z_coverage.py ${1:-100} ${2:-24} 2>&1 | grep "^-z" | tail -6
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-70s calls %5s total ms %8.1f avg us %9.1f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT
