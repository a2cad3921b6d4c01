#!/bin/bash
# the match kernels of one pass in launch order (grid, LDS, duration): which levels the serial replays cost -- tools/r6_match_trace.sh [bench flags]
set -u
R=$(pwd); OUT=$R/gpurun_out/r6_match_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli "$@" > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "kc_pack2bit_kernel" in r["Kernel_Name"])
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0]
    if "match" not in n: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d < 20: continue
    print("%-28s grid %8s wg %4s lds %6s  %8.1f us" % (n[-28:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"), d))
PY
rm -f $OUT/*/*_kernel_trace.csv
