#!/usr/bin/env python3
"""Turns the raw output of tools/collect_profile.sh into the small, committed files under profiles/:
   <tag>_kernel_stats.csv (rocprofv3 --stats), <tag>_bench_line.json, <tag>_traffic.json (per-kernel HBM bytes per launch).

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of a wide coalesced streaming read (calibrated here on kc_pack2bit_kernel, whose read volume is known:
it reads every input byte once with 16-B loads), so reads are doubled; WRITE_SIZE is taken as is.
Round 6 (profiles/r06_fetch_calibration.txt, tools/fetch_calib.hip): FETCH_SIZE is TCC_EA0_RDREQ x 64 B for every access class, and
every request is one 128-byte line, scattered or streamed -- so the x2 gives the bytes of the lines requested for scattered lookups too,
and `fetch_requests` (raw / 64) is the number to hold against the ~50 G requests/s this GPU sustains for either class."""
import csv
import glob
import json
import re
import shutil
import sys


def main(src, tag):
    stats = glob.glob(src + "/stats/*/*_kernel_stats.csv")[0]
    shutil.copy(stats, "profiles/%s_kernel_stats.csv" % tag)
    line = [l for l in open(src + "/bench_stats.log").read().splitlines() if l.startswith('{"metric"')][-1]
    bench = json.loads(line)
    json.dump(bench, open("profiles/%s_bench_line.json" % tag, "w"), indent=1)
    agg = {}
    for sub in ("pmc_fetch", "pmc_write"):
        for r in csv.DictReader(open(glob.glob(src + "/%s/*/*_counter_collection.csv" % sub)[0])):
            k = r["Kernel_Name"].split("(")[0]
            a = agg.setdefault(k, {"launches": set()})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            a["launches"].add((sub, r["Dispatch_Id"]))
    out = {}
    nbytes = None
    for k, a in agg.items():
        n = max(1, len([1 for s, _ in a["launches"] if s == "pmc_fetch"]))
        fetch, write = a.get("FETCH_SIZE", 0.0) * 1024 / n, a.get("WRITE_SIZE", 0.0) * 1024 / n
        hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
        out[k] = {"launches_per_pass": n, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
                  "hbm_bytes_per_launch": 2 * fetch + write, "fetch_requests": fetch / 64, "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
    pk = [k for k in out if "kc_pack2bit_kernel" in k]
    if pk:
        # calibration: the pack kernel reads `len` input bytes exactly once
        cfg = bench["config"]
        # (records x (bases + the separator): read off the workload string, so that k = 63 / 127 -- 1 062 / 1 126 bases per record -- calibrate right)
        mw = re.search(r"(\d+) records x (\d+) bp", cfg.get("workload", ""))
        nbytes = int(mw.group(1)) * (int(mw.group(2)) + 1) if mw else None
        out["_calibration"] = {"kernel": pk[0], "known_read_bytes": nbytes, "fetch_size_bytes_raw": out[pk[0]]["fetch_size_bytes_raw"],
                               "ratio_known_over_raw": round(nbytes / out[pk[0]]["fetch_size_bytes_raw"], 3) if nbytes else None}
    json.dump(out, open("profiles/%s_traffic.json" % tag, "w"), indent=1, sort_keys=True)
    dom = [k for k in out if "kc_probe_block_kernel" in k]
    if dom:
        print(dom[0], out[dom[0]])
    print(out.get("_calibration"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
