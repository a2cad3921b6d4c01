#!/usr/bin/env python3
"""Where one pass is idle: tools/r6_gaps.py <dir with *_kernel_trace.csv> -- the LAST pass of the trace (from the last pack kernel on):
wall, busy, and the idle gaps of 15 us or more with the kernels on either side (a gap = a host round trip or launch latency)."""
import csv, glob, sys, os, collections
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-50:]))
rows.sort()
last = max(i for i, r in enumerate(rows) if "kc_pack2bit_kernel" in r[2])
ph = rows[last:]
wall = ph[-1][1] - ph[0][0]
busy = 0; cur = ph[0][0]; gaps = []
for i, (s, e, n) in enumerate(ph):
    if s > cur and i: gaps.append((s - cur, ph[i - 1][2], n, (cur - ph[0][0]) / 1e6))
    if e > cur:
        busy += e - max(s, cur); cur = e
print("last pass: %d launches, wall %.3f ms, busy %.3f ms, idle %.3f ms" % (len(ph), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
big = [g for g in gaps if g[0] >= 15000]
print("%d gaps >= 15 us, %.3f ms in all; gaps < 15 us: %d, %.3f ms" % (len(big), sum(g[0] for g in big) / 1e6, len(gaps) - len(big), sum(g[0] for g in gaps if g[0] < 15000) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for g in big:
    k = g[1] + " -> " + g[2]; agg[k][0] += g[0]; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:30]:
    print("  %7.1f us in %3d gap(s)  %s" % (t / 1e3, c, k))
