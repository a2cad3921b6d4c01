"""round 5: a genome-like input with what real genomes add to random sequence -- satellite arrays (a 171 bp unit x hundreds of copies with
a few % divergence), microsatellites (units of 1-6 bp), poly-A runs: skewed minimizers (overfull table blocks, second homes), huge
multiplicities, cycles.  Size-independent checks + time.  python tools/r5_repeats_shape.py [Gbp] [satellite fraction]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kmercamel_amd as kc
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sat = float(sys.argv[2]) if len(sys.argv) > 2 else 0.08
rng = np.random.default_rng(11)
nrec, L = 10, int(gbp * 1e8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
recs = [rng.integers(0, 4, size=L, dtype=np.uint8) for _ in range(nrec)]
total = nrec * L
n_arr = int(total * sat / (171 * 300))
for _ in range(n_arr):   # satellite arrays: one of 5 families, copies diverge by 3 %
    fam = np.random.default_rng(int(rng.integers(0, 5)) + 1000).integers(0, 4, size=171, dtype=np.uint8)
    copies = int(rng.integers(100, 500))
    arr = np.tile(fam, copies)
    mut = rng.random(arr.size) < 0.03
    arr[mut] = (arr[mut] + rng.integers(1, 4, size=int(mut.sum()), dtype=np.uint8)) & 3
    r = int(rng.integers(0, nrec)); p = int(rng.integers(0, L - arr.size))
    recs[r][p:p + arr.size] = arr
for _ in range(int(total / 2e4)):   # microsatellites and poly-A
    unit = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
    ln = int(rng.integers(60, 600))
    arr = np.tile(unit, ln // len(unit) + 1)[:ln]
    r = int(rng.integers(0, nrec)); p = int(rng.integers(0, L - ln))
    recs[r][p:p + ln] = arr
seq = np.concatenate([np.concatenate([acgt[r], np.array([10], dtype=np.uint8)]) for r in recs])[:-1]
del recs
ctx = kc.Context(0)
k = 31
for i in range(3):
    t0 = time.time()
    r = kc.compute(ctx, seq, k, True)
    dt = time.time() - t0
    st = r.stats
    if i == 0:
        o = r.array().copy()
    else:
        assert np.array_equal(o, r.array()), "two passes differ"
    r.close()
    print("pass %d: N=%d S=%d L=%d rounds=%d cycles=%d | %.1f ms incl. H2D; device ms: insert %.1f simplitigs %.1f overlap %.1f emit %.1f" % (
        i, st["n_kmers"], st["n_simplitigs"], st["length"], st["matching_rounds"], st["cycles_cut"], dt * 1e3,
        st["us_insert_phase"] / 1e3, st["us_simplitig_phase"] / 1e3, st["us_overlap_phase"] / 1e3, st["us_emit_phase"] / 1e3), flush=True)
N = st["n_kmers"]
assert int(np.count_nonzero(o < 97)) == N, "ones != N"
a = kc.KmerSet(ctx, k, True, capacity_hint=len(seq)); a.AddKMers(seq); assert a.size() == N; a.close()
b = kc.KmerSet(ctx, k, True, capacity_hint=len(o)); b.AddKMers(o, case_sensitive=True); assert b.size() == N, (b.size(), N); b.close()
print("checks ok: ones == N, both sets of size N, two more passes byte-identical")
ctx.close()
