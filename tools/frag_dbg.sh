cd /tmp && export TMPDIR=/tmp
export KC_FRAG=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf0 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cli --no-cpu-baseline > /tmp/line.txt 2>&1
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/pf0/*/*kernel_stats.csv")[0])):
    if "kc_frag_contract" in r["Name"] or "walk_measure" in r["Name"] or "stream_emit" in r["Name"]:
        print(r["Name"][:45], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
tail -1 /tmp/line.txt | cut -c1-900
cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fragment or alternative" 2>&1 | tail -3
