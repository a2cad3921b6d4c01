#!/bin/bash
# cost of the neighbour-probe kernel at BASELINE configs[1] (10^8 31-mers): time per launch (rocprofv3 stats) and the dynamic
# instruction mix per wave (one PMC pass).  Usage (through gpurun, repo root): bash tools/probe_kernel_cost.sh
R=$(pwd); OUT=$R/gpurun_out/probe_cost; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--records 100000 --seed 1 --no-cpu-baseline --no-cli"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS --steps 5 --warmup 1 > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:56].ljust(56), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3))
f = glob.glob(sys.argv[1] + "/pmc/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
for k, d in agg.items():
    if not any(t in k for t in ("choose_dense", "probe_block", "frag_contract", "walk_measure", "bucket_scatter", "block_build")): continue
    w = d["SQ_WAVES"]
    big = max(1, sum(1 for _ in disp[k]) if "block_build" not in k else len(disp[k]) // 2)   # launches that process the whole set (the build has a sample launch too)
    nk = 1e8 * big
    print("  %d launch(es); wave-instructions x 64 per k-mer: VALU %.0f SALU %.0f LDS %.1f VMEM_RD %.2f" % (len(disp[k]), d["SQ_INSTS_VALU"] * 64 / nk, d["SQ_INSTS_SALU"] * 64 / nk, d["SQ_INSTS_LDS"] * 64 / nk, d["SQ_INSTS_VMEM_RD"] * 64 / nk))
    print(k.ljust(40), "waves %d per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f | wave cycles %.0f, wait %.2f, VALU active %.2f" % (
        w, d["SQ_INSTS_VALU"] / w, d["SQ_INSTS_SALU"] / w, d["SQ_INSTS_LDS"] / w, d["SQ_INSTS_VMEM_RD"] / w, d["SQ_WAVE_CYCLES"] / w,
        d["SQ_WAIT_ANY"] / max(1, d["SQ_WAVE_CYCLES"]), d["SQ_ACTIVE_INST_VALU"] / max(1, d["SQ_WAVE_CYCLES"])))
PY
rm -f $OUT/*/*/*_kernel_trace.csv
