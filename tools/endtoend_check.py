#!/usr/bin/env python3
"""GPU-box check of (1) the CLI's end-to-end wall time on the cfg[1] FASTA (host read + H2D + kernels + D2H + write) and
(2) the single-long-chain worst case of the per-chain walks."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gen
BIN = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
d = "/dev/shm/kc_e2e"
os.makedirs(d, exist_ok=True)
recs = gen.random_records(100000, 1030, 1)
fa = os.path.join(d, "cfg2.fa")
with open(fa, "wb") as f:
    for i, r in enumerate(recs):
        f.write(b">r%d\n%s\n" % (i, r))
for rep in range(2):
    t = time.time()
    p = subprocess.run([BIN, "compute", "-k", "31", "-o", os.path.join(d, "out.fa"), fa], capture_output=True, text=True)
    dt = time.time() - t
    print("CLI cfg2 run %d: rc=%d wall=%.2fs -> %.1f Mkmers/s end-to-end" % (rep, p.returncode, dt, 100.0 / dt))
print(p.stderr.strip().splitlines()[-1])
import kmercamel_amd as kc
ctx = kc.Context(0)
for L in (1000000, 5000000):
    seq = gen.random_records(1, L, 3)[0]
    t = time.time()
    r = kc.compute(ctx, seq, 31, True)
    dt = time.time() - t
    print("single record of %d bp: S=%d N=%d in %.2fs (device us: insert %d simplitigs %d overlap %d emit %d)" % (
        L, r.stats["n_simplitigs"], r.stats["n_kmers"], dt, r.stats["us_insert_phase"], r.stats["us_simplitig_phase"], r.stats["us_overlap_phase"], r.stats["us_emit_phase"]))
    r.close()
ctx.close()
