#!/usr/bin/env python3
"""Real-data SHAPES at scale (the benchmark's i.i.d. 1 kbp records have S = R, no branching and no long chains; the reference
treats all shapes alike, simplitigs.h:142-189):

  genome    R records x L bp of random sequence with planted repeats -- dispersed copies of 2-10 kbp segments on either strand
            (branching), tandem repeats (cycles in the de Bruijn graph), a few duplicated stretches: long chains, the
            segment-parallel walks, the cycle rule.  Default 25 x 10^8 bp = 2.5 G k-mers.
  coverage  reads of RL bp at COV-fold coverage of a random genome from both strands with substitution errors: every true
            k-mer occurs ~COV times, every error adds up to k new k-mers (tips, bubbles), the table is sized by occurrences.
            Default 30x of 10^8 bp with 1 % errors: 3 10^9 occurrences, ~10^9 distinct k-mers.

`python tools/shapes.py genome|coverage [--scale F] [--check] [--passes N]`: generates the input on the host (numpy, seeded),
runs `compute -k 31` on the device and prints one JSON line {shape, N, S, L, ms per pass, phases, Mk-mers/s, checks}.
--check verifies size-independent properties without the oracle: ones == N (min-one mask), the k-mer set read back off the
output equals the input's set in size, two passes give identical bytes.  Used by tests/test_gpu_shapes.py at reduced scale."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def genome_records(n_records, length, seed, repeats_per_mbp=4.0, tandems_per_mbp=2.0):
    """list of uint8 code arrays (0..3).  Per Mbp: `repeats_per_mbp` dispersed copies of a 2-10 kbp segment of the same record set
    (half of them reverse-complemented) and `tandems_per_mbp` tandem repeats (unit 40-400 bp, 3-8 copies)."""
    rng = np.random.default_rng([seed, 1])
    recs = []
    for r in range(n_records):
        recs.append(np.random.default_rng([seed, 2, r]).integers(0, 4, size=length, dtype=np.uint8))
    total = n_records * length
    n_rep = int(total / 1e6 * repeats_per_mbp)
    for _ in range(n_rep):
        ln = int(rng.integers(2000, 10000))
        if ln >= length:
            continue
        a, b = int(rng.integers(0, n_records)), int(rng.integers(0, n_records))
        pa, pb = int(rng.integers(0, length - ln)), int(rng.integers(0, length - ln))
        seg = recs[a][pa:pa + ln].copy()
        if rng.random() < 0.5:
            seg = (3 - seg)[::-1]
        recs[b][pb:pb + ln] = seg
    n_tan = int(total / 1e6 * tandems_per_mbp)
    for _ in range(n_tan):
        unit = int(rng.integers(40, 400))
        copies = int(rng.integers(3, 9))
        if unit * copies >= length:
            continue
        b = int(rng.integers(0, n_records))
        pb = int(rng.integers(0, length - unit * copies))
        u = recs[b][pb:pb + unit].copy()
        recs[b][pb:pb + unit * copies] = np.tile(u, copies)
    return recs


def join_records(code_recs):
    """ASCII records joined by '\\n' (one uint8 array)"""
    total = sum(len(r) + 1 for r in code_recs)
    out = np.empty(total, dtype=np.uint8)
    o = 0
    for r in code_recs:
        out[o:o + len(r)] = ACGT[r]
        out[o + len(r)] = 10
        o += len(r) + 1
    return out[:-1]


def coverage_reads(genome_len, coverage, read_len, error_rate, seed):
    rng = np.random.default_rng([seed, 3])
    genome = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
    n_reads = int(genome_len * coverage / read_len)
    a = np.empty((n_reads, read_len + 1), dtype=np.uint8)
    for lo in range(0, n_reads, 500000):   # in slices: the index matrix of all reads at once would take tens of GB
        hi = min(n_reads, lo + 500000)
        pos = rng.integers(0, genome_len - read_len, size=hi - lo)
        codes = genome[pos[:, None] + np.arange(read_len)[None, :]]
        rev = rng.random(hi - lo) < 0.5
        codes[rev] = 3 - codes[rev][:, ::-1]
        err = rng.random(codes.shape) < error_rate
        codes[err] = (codes[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
        a[lo:hi, :read_len] = ACGT[codes]
    a[:, read_len] = 10
    return a.reshape(-1)[:-1]


def make_input(shape, scale, seed=11):
    if shape == "genome":
        n_records, length = 25, max(40000, int(1e8 * scale))
        return join_records(genome_records(n_records, length, seed)), {"records": n_records, "record_bp": length}
    if shape == "coverage":
        g = max(100000, int(1e8 * scale))
        return coverage_reads(g, 30, 150, 0.01, seed), {"genome_bp": g, "coverage": 30, "read_bp": 150, "error_rate": 0.01}
    raise SystemExit("shape must be genome or coverage")


def run(shape, scale, passes=3, check=False, k=31, device=0, env=None):
    import kmercamel_amd as kc
    t0 = time.time()
    seq, cfg = make_input(shape, scale)
    t_gen = time.time() - t0
    ctx = kc.Context(device)
    first = None
    times, st = [], None
    for i in range(passes):
        t0 = time.time()
        r = kc.compute(ctx, seq, k, True)
        times.append(time.time() - t0)
        st = r.stats
        if check:
            a = r.array().copy()
            if first is None:
                first = a
            else:
                assert np.array_equal(a, first), "two passes differ"
        r.close()
    N, S, L = st["n_kmers"], st["n_simplitigs"], st["length"]
    checks = {}
    if check:
        assert int(np.count_nonzero(first < 97)) == N, "mask is not min-one: %d ones, %d k-mers" % (int(np.count_nonzero(first < 97)), N)
        assert N + k - 1 <= L <= N + S * (k - 1)
        b = kc.KmerSet(ctx, k, True, capacity_hint=L)
        b.AddKMers(first.tobytes(), case_sensitive=True)
        assert b.size() == N, "round trip: the output represents %d k-mers, the input has %d" % (b.size(), N)
        b.close()
        checks = {"ones_eq_N": True, "set_round_trip": True, "deterministic": passes > 1}
    dev_ms = st["us_total"] / 1e3
    line = {"shape": shape, "config": cfg, "k": k, "n_kmers": N, "n_simplitigs": S, "superstring_length": L,
            "ms_per_pass_device": round(dev_ms, 3), "Mkmers_per_s_device": round(N / dev_ms / 1e3, 1),
            "phases_ms": {p: round(st[p] / 1e3, 3) for p in ("us_insert_phase", "us_simplitig_phase", "us_overlap_phase", "us_emit_phase")},
            "matching_rounds": st["matching_rounds"], "cycles_cut": st.get("cycles_cut", 0), "overlap_repairs": st.get("overlap_repairs", 0),
            "wall_ms_incl_h2d": [round(t * 1e3, 1) for t in times], "input_bytes": int(seq.size), "generate_s": round(t_gen, 1), "checks": checks}
    ctx.close()
    return line


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("shape")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    print(json.dumps(run(args.shape, args.scale, args.passes, args.check)), flush=True)
