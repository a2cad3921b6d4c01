#!/bin/bash
# which device-to-device copies of a pass are long?  kernel trace of one bench pass; copyBuffer dispatches over 50 us with their neighbours
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5/copies; ARGS="$@"; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli $ARGS > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last pack2bit on
last = max(i for i, r in enumerate(rows) if "pack2bit" in r["Kernel_Name"])
rows = rows[last:]
tot = 0
for i, r in enumerate(rows):
    if "copyBuffer" not in r["Kernel_Name"]: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d > 50:
        prev = next((rows[j]["Kernel_Name"][:50] for j in range(i - 1, -1, -1) if "copyBuffer" not in rows[j]["Kernel_Name"] and "fillBuffer" not in rows[j]["Kernel_Name"]), "")
        nxt = next((rows[j]["Kernel_Name"][:50] for j in range(i + 1, len(rows)) if "copyBuffer" not in rows[j]["Kernel_Name"] and "fillBuffer" not in rows[j]["Kernel_Name"]), "")
        print("%8.1f us  grid %s  after [%s]  before [%s]" % (d, r.get("Grid_Size_X", r.get("Grid_Size", "?")), prev, nxt))
print("copies of the pass: %.2f ms" % (tot / 1e3))
PY
rm -rf $OUT
