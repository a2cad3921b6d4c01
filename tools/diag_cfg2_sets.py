"""diagnostic: at 2.5 G k-mers, is the k-mer set read back off the output equal to the input's set? (full membership, both ways)"""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import kmercamel_amd as kc

records = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
k = 31
ctx = kc.Context(0)
seq = bench.generate(2, 0, records, 1030)
r = kc.compute(ctx, seq, k, True)
N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
o = r.array().copy(); r.close()
print("N", N, "S", S, "L", L, "ones", int(np.count_nonzero(o < 97)), flush=True)

def export(s):
    n = s.size()
    buf = np.empty(max(1, n), dtype=np.uint64)
    s.ctx.check(s.ctx.lib.kc_set_export(s.h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    return buf[:n]

def digest(keys):
    mix = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        return len(keys), int(keys.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(keys * mix))

def misses(s, keys, step=100_000_000):
    miss = 0; ex = []
    for i in range(0, len(keys), step):
        chunk = np.ascontiguousarray(keys[i:i + step])
        out = np.empty(len(chunk), dtype=np.uint8)
        s.ctx.check(s.ctx.lib.kc_set_contains(s.h, chunk.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(chunk), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))))
        bad = np.flatnonzero(out == 0)
        miss += len(bad)
        ex.extend(int(x) for x in chunk[bad[:5]])
    return miss, ex[:10]

a = kc.KmerSet(ctx, k, True, capacity_hint=len(seq)); a.AddKMers(seq)
ka = export(a); print("a", digest(ka), flush=True)
ka2 = export(a); print("a again", digest(ka2), "same multiset order-free:", digest(ka) == digest(ka2), flush=True)
del ka2
a.close()
b = kc.KmerSet(ctx, k, True, capacity_hint=L); b.AddKMers(o, case_sensitive=True)
kb = export(b); print("b", digest(kb), flush=True)
print("keys of a missing in b:", misses(b, ka), flush=True)
b.close()
a = kc.KmerSet(ctx, k, True, capacity_hint=len(seq)); a.AddKMers(seq)
print("keys of b missing in a:", misses(a, kb), flush=True)
print("dups in ka sample? unique of first 50M:", len(np.unique(ka[:50_000_000])), flush=True)
a.close()
