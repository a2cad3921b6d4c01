#!/usr/bin/env python3
"""Timeline of phase 2 out of a rocprofv3 kernel trace: tools/ov_timeline.py <dir with *_kernel_trace.csv>
Takes the LAST pass in the trace (from the last kc_probe_block_kernel on), cuts out the stretch from the first ov_* / ovg_* kernel to
the first em_* kernel and prints: wall, busy, idle, launches, per-kernel totals, and the copy / fill kernels with their durations."""
import csv, glob, sys, collections, os
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last_probe = max(i for i, r in enumerate(rows) if "kc_probe_block_kernel" in r[2])
tail = rows[last_probe:]
i0 = next(i for i, r in enumerate(tail) if r[2].startswith("ov_") or r[2].startswith("ovg_") or " ov_" in r[2] or " ovg_" in r[2] or "kc_end_kmers" in r[2])
i1 = next((i for i, r in enumerate(tail) if i > i0 and ("em_" in r[2].split("(")[0])), len(tail))
ph = tail[i0:i1]
wall = ph[-1][1] - ph[0][0]
busy = 0; cur_end = ph[0][0]
for s, e, n in ph:
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print("phase 2: %d launches, wall %.3f ms, busy %.3f ms, idle %.3f ms" % (len(ph), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in ph:
    k = n.split("(")[0][-60:]
    tot[k][0] += e - s; tot[k][1] += 1
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %-62s %5d calls %9.3f ms" % (k, c, t / 1e6))
whole = tail
print("whole pass after the probe kernel: %d launches" % len(whole))
allp = rows[max(0, last_probe - 40):]
cp = [(e - s) / 1e3 for s, e, n in allp if "copyBuffer" in n]
print("copyBuffer in the last pass (approx.): %d, durations us: %s" % (len(cp), " ".join("%.0f" % x for x in sorted(cp, reverse=True)[:40])))
