"""The 'next' rows of SURVEY 8f at the benchmark's size (round 5): compute -M, -z 2, maskopt max-one / min-one, lowerbound on the cfg[2]
record set (or N records): times and size-independent checks.  python tools/r5_next_rows_at_scale.py [records]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import kmercamel_amd as kc

records = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
k = 31
seq = bench.generate(2, 0, records, 1030)
ctx = kc.Context(0)

def timed(what, f, reps=2):
    out = None
    for i in range(reps):
        t0 = time.time(); out = f(); dt = time.time() - t0
        print("%-44s run %d: %8.1f ms" % (what, i, dt * 1e3), flush=True)
        if i + 1 < reps and hasattr(out, "close"): out.close()
    return out

r = timed("compute (host sequence in, no mask)", lambda: kc.compute(ctx, seq, k, True))
N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
o = r.array().copy(); r.close()
rm = timed("compute -M", lambda: kc.compute(ctx, seq, k, True, want_mask=True))
st = rm.stats
out, mask = rm.strings(want_mask=True) if L < (1 << 31) else (None, None)
print("   phases ms: insert %.1f simplitigs %.1f overlap %.1f emit+mask %.1f" % tuple(st[p] / 1e3 for p in ("us_insert_phase", "us_simplitig_phase", "us_overlap_phase", "us_emit_phase")))
rm.close()
if mask is not None:
    m = np.frombuffer(mask.encode() if isinstance(mask, str) else mask, dtype=np.uint8)
    ones_max = int(np.count_nonzero(m < 97))
    assert ones_max >= N, (ones_max, N)
    assert (m | 32 == o | 32).all(), "-M changes characters"
    print("   -M: ones %d >= N %d, characters kept" % (ones_max, N))
    del m
rz = timed("compute -z 2 (every k-mer occurs once: empty)", lambda: None if False else _z())  if False else None
def _z():
    try:
        x = kc.compute(ctx, seq, k, True, min_frequency=2)
        return x
    except kc.KcError as e:
        return e
t0 = time.time(); z = _z(); print("%-44s %8.1f ms -> %s" % ("compute -z 2 (i.i.d. reads: no k-mer twice)", (time.time() - t0) * 1e3, ("N=%d" % z.stats["n_kmers"]) if hasattr(z, "stats") else repr(z)[:80]), flush=True)
if hasattr(z, "close"): z.close()
ms = bytes(o)
for alg in ("max-one", "min-one"):
    t0 = time.time(); res = kc.Optimize(ctx, ms, k, True, alg); dt = time.time() - t0
    a = np.frombuffer(res.encode() if isinstance(res, str) else res, dtype=np.uint8)
    print("%-44s %8.1f ms  ones %d" % ("maskopt " + alg + " (host string in and out)", dt * 1e3, int(np.count_nonzero(a < 97))), flush=True)
    assert (a | 32 == o | 32).all()
    if alg == "min-one": assert int(np.count_nonzero(a < 97)) == N
s = kc.KmerSet(ctx, k, True, capacity_hint=len(seq)); s.AddKMers(seq)
sim = kc.get_simplitigs(s)
t0 = time.time(); lb = kc.LowerBoundLength(sim, True); dt = time.time() - t0
print("%-44s %8.1f ms  -> %d (superstring %d)" % ("lowerbound on the simplitigs", dt * 1e3, lb, L), flush=True)
assert lb <= L
sim.close(); s.close(); ctx.close()
print("ok")
