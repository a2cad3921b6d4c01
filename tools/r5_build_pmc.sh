#!/bin/bash
# PMC passes over the set build alone (tools/r5_build_ab.py): tools/r5_build_pmc.sh records "VARIANT" "counters pass 1" ["counters pass 2" ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r5/bpmc; rm -rf $OUT; mkdir -p $OUT
rec=$1; var=$2; shift 2
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p$i -o t -- python3 $R/tools/r5_build_ab.py $rec "$var" > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log | cut -c1-200
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        if not any(t in k for t in ("bucket_scatter", "block_build")): continue
        n = len(disp[k])
        print(k, "launches", n, " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(d.items())))
PY
rm -rf $OUT/p*/
