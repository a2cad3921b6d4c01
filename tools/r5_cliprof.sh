#!/bin/bash
# round 5: kernel trace of the CLI at cfg[2] (tokeniser kernels, gaps)
mkdir -p gpurun_out/r5
python3 - > gpurun_out/r5/gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
KC_CLI_NO_FAST_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/r5/cli_prof -- $R/kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_r5/out.txt /dev/shm/kc_r5/in.fa > $R/gpurun_out/r5/cli_prof.log 2>&1
cd $R
rm -rf /dev/shm/kc_r5
python3 - <<'PY'
import csv, glob
for pat in ("*kernel_stats.csv", "*memory_copy_stats.csv"):
    for f in glob.glob("gpurun_out/r5/cli_prof/*/" + pat):
        for r in list(csv.DictReader(open(f)))[:14]:
            print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%10.1f us avg" % (float(r["AverageNs"]) / 1e3), "%10.1f ms total" % (float(r["TotalDurationNs"]) / 1e6))
# timeline of the first 40 tokeniser kernels + copies
rows = []
for f in glob.glob("gpurun_out/r5/cli_prof/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob("gpurun_out/r5/cli_prof/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") ))
rows.sort()
t0 = rows[0][0]
for s, e, n in rows[:60]:
    print("%9.3f ms  +%8.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, n))
print("last event ends at %.3f ms" % ((rows[-1][1] - t0) / 1e6))
PY
find gpurun_out/r5/cli_prof -name "*_trace.csv" -delete
rm -rf gpurun_out/r5/cli_prof
