#!/usr/bin/env python3
"""Timing on the shape real sequencing data has (not the benchmark's i.i.d. reads): coverage of a random genome by 150 bp reads
from both strands with substitution errors.  `python tools/time_coverage_reads.py [genome_Mbp] [coverage] [error_rate]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmercamel_amd as kc

gm = float(sys.argv[1]) if len(sys.argv) > 1 else 50
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 10
er = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
rng = np.random.default_rng(7)
G, rl, k = int(gm * 1e6), 150, 31
n_reads = int(G * cov / rl)
genome = rng.integers(0, 4, size=G, dtype=np.uint8)
a = np.empty((n_reads, rl + 1), dtype=np.uint8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for lo in range(0, n_reads, 500000):   # in slices: the index matrix of all reads at once would take tens of GB
    hi = min(n_reads, lo + 500000)
    pos = rng.integers(0, G - rl, size=hi - lo)
    codes = genome[pos[:, None] + np.arange(rl)[None, :]]
    rev = rng.random(hi - lo) < 0.5
    codes[rev] = 3 - codes[rev][:, ::-1]
    err = rng.random(codes.shape) < er
    codes[err] = (codes[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
    a[lo:hi, :rl] = acgt[codes]
a[:, rl] = 10
seq = a.reshape(-1)[:-1]
ctx = kc.Context(0)
for i in range(3):
    t0 = time.time()
    r = kc.compute(ctx, seq, k, True)
    dt = time.time() - t0
    st = r.stats
    r.close()
    print("%d reads x %d bp (%.0f Mbp genome, %.0fx, %.1f %% errors): N=%d S=%d L=%d rounds=%d cycles=%d repairs=%d | %.1f ms incl. H2D; device ms: insert %.1f simplitigs %.1f overlap %.1f emit %.1f" % (
        n_reads, rl, gm, cov, er * 100, st["n_kmers"], st["n_simplitigs"], st["length"], st["matching_rounds"], st["cycles_cut"], st.get("overlap_repairs", -1), dt * 1e3,
        st["us_insert_phase"] / 1e3, st["us_simplitig_phase"] / 1e3, st["us_overlap_phase"] / 1e3, st["us_emit_phase"] / 1e3), flush=True)
ctx.close()
