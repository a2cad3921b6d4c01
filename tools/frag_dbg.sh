cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf0 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cli --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/pf0/*/*kernel_stats.csv")[0])):
    if "kc_frag_contract" in r["Name"] or "walk_measure" in r["Name"] or "choose_dense" in r["Name"] or "stream_emit" in r["Name"]:
        print(r["Name"][:45], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
