"""round 5: `compute -z 2` on reads with coverage (the filter real read sets are run with): python tools/r5_z_coverage.py [genome_Mbp] [coverage]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmercamel_amd as kc
gm = float(sys.argv[1]) if len(sys.argv) > 1 else 50
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(7)
G, rl, k, er = int(gm * 1e6), 150, 31, 0.01
n_reads = int(G * cov / rl)
genome = rng.integers(0, 4, size=G, dtype=np.uint8)
a = np.empty((n_reads, rl + 1), dtype=np.uint8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
for lo in range(0, n_reads, 500000):
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
for z in (1, 2, 3):
    for i in range(2):
        t0 = time.time()
        r = kc.compute(ctx, seq, k, True, min_frequency=z)
        dt = time.time() - t0
        st = r.stats
        r.close()
        print("-z %d: N=%d S=%d L=%d | %.1f ms incl. H2D; device ms: insert %.1f simplitigs %.1f overlap %.1f emit %.1f" % (
            z, st["n_kmers"], st["n_simplitigs"], st["length"], dt * 1e3, st["us_insert_phase"] / 1e3, st["us_simplitig_phase"] / 1e3, st["us_overlap_phase"] / 1e3, st["us_emit_phase"] / 1e3), flush=True)
ctx.close()
