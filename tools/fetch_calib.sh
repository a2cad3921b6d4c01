#!/bin/bash
# FETCH_SIZE calibration per access class (through gpurun, repo root): bash tools/fetch_calib.sh [footprint GB]
# One rocprofv3 pass per counter group (FETCH_SIZE takes 3 of the 4 TCC slots); the program after `--` is the binary itself.
R=$(pwd); OUT=$R/gpurun_out/fetch_calib; rm -rf $OUT; mkdir -p $OUT
GB=${1:-40}
hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o $R/tools/fetch_calib.bin || exit 1
cd /tmp && export TMPDIR=/tmp
$R/tools/fetch_calib.bin $GB > $OUT/plain.txt 2>&1
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
i=0
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p$i -- $R/tools/fetch_calib.bin $GB > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("calib_"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    print(k, {c: agg[k][c] / n[k][c] for c in sorted(agg[k])})
PY
rm -f $OUT/*/*/*_kernel_trace.csv $OUT/*/*/*agent_info.csv
