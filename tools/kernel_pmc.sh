#!/bin/bash
# PMC counters of one kernel at BASELINE configs[1] (10^8 31-mers), one rocprofv3 pass per counter group.
# Usage (through gpurun, repo root): bash tools/kernel_pmc.sh <kernel-name-substring> "<counters of pass 1>" ["<counters of pass 2>" ...]
# Environment: KC_PMC_ARGS = bench.py arguments (default: configs[1], one step)
R=$(pwd); OUT=$R/gpurun_out/kernel_pmc; rm -rf $OUT; mkdir -p $OUT
KERN=$1; shift
ARGS=${KC_PMC_ARGS:---records 100000 --seed 1 --no-cpu-baseline --no-cli --steps 1 --warmup 0}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p$i -- python3 $R/bench.py $ARGS > /dev/null 2>&1
done
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(agg):
    print("%-32s total %.4g  per launch %.4g  (%d launches)" % (k, agg[k], agg[k] / max(1, n[k]), n[k]))
PY
rm -f $OUT/*/*/*_kernel_trace.csv
