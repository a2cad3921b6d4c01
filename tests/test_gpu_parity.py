"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bar: bit-exact everywhere (integer / byte / index work).  Tiers:
  T-A  byte-exact vs the reference semantics: primitives, set contents, C1 edge arrays, C2 bytes, Global, sparse branch,
       `compute -S`, lower bound, maskopt.
  T-B  raw-FASTA compute: identical N and k-mer set, ones == N, -M == maskopt max-one of -o, L reported vs reference L.
  T-C  raw-FASTA compute byte-identical to the oracle's canonical-order mode.
"""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import gen
import oracle_lib
import kmercamel_amd as kc

pytestmark = pytest.mark.gpu

NONE = (1 << 64) - 1
KS = [2, 5, 13, 31, 32, 47, 63, 64, 100, 127]


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


# ------------------------------------------------------------------------------------------------ A2/A6: set contents
@pytest.mark.parametrize("k", KS)
@pytest.mark.parametrize("complements", [True, False])
def test_set_contents_match_oracle(ctx, oracle, k, complements):
    rng = random.Random(1000 + k)
    for case_sensitive in (False, True):
        parts = []
        for _ in range(30):
            s = _rand_seq(rng, rng.choice([0, 1, k - 1, k, k + 1, 3 * k, 500]), "ACGTacgtN" if rng.random() < 0.5 else "ACGT")
            parts.append(s)
        # AddKMers is per record (parser.h:22-49): feed the records one by one AND joined by a separator
        seq = "\n".join(parts)
        want = sorted(set(x for p in parts for x in oracle.kmer_set(p, k, complements, case_sensitive)))
        s1 = kc.KmerSet(ctx, k, complements, capacity_hint=len(seq))
        s1.AddKMers(seq, case_sensitive)
        assert s1.size() == len(want)
        assert sorted(s1.keys()) == want
        s2 = kc.KmerSet(ctx, k, complements)   # grows on demand
        for p in parts:
            s2.AddKMers(p, case_sensitive)
        assert sorted(s2.keys()) == want
        probe = want[:50] + [(x + 1) & ((1 << (2 * k)) - 1) for x in want[:50]]
        wset = set(want)
        assert s1.contains(probe) == [p in wset for p in probe]
        s1.close(); s2.close()


def test_device_input_at_any_alignment(ctx, tmp_path):
    """kc_compute with the input already on the device (KC_FLAG_INPUT_ON_DEVICE), at pointers that are not 16-byte aligned: the 2-bit
    packing reads 16-byte pieces where it may and single bytes where it must.  The buffer comes from kc_read_fasta_device; computing from
    `shift` bytes into it is computing on the joined records without their first `shift` characters.  Mixed case, N, ragged records, an
    input that ends inside a piece."""
    import ctypes
    rng = random.Random(77)
    for k in (5, 31, 40):
        parts = [_rand_seq(rng, rng.choice([k - 1, k, k + 1, 3 * k, 200, 1030]), "ACGTacgtN" if rng.random() < 0.3 else "ACGT") for _ in range(60)]
        path = str(tmp_path / ("in%d.fa" % k))
        with open(path, "w") as f:
            for i, p_ in enumerate(parts):
                f.write(">r%d\n%s\n" % (i, p_))
        joined = "\n".join(parts).encode()
        d, n, fq = ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_int(0)
        ctx.check(ctx.lib.kc_read_fasta_device(ctx.h, path.encode(), ctypes.byref(d), ctypes.byref(n), ctypes.byref(fq)))
        assert n.value == len(joined) + 1 and not fq.value   # (every record is preceded by a separator: where its header line was)
        content = b"\n" + joined
        try:
            for shift in (0, 1, 3, 8, 15, 16, 21):
                want = kc.compute(ctx, content[shift:], k, True)
                ref = want.array().copy()
                want.close()
                r = kc.compute(ctx, None, k, True, device_ptr=d.value + shift, length=n.value - shift)
                same = np.array_equal(r.array(), ref)
                r.close()
                assert same, (k, shift)
        finally:
            ctx.lib.kc_device_free(ctx.h, d)


def test_read_kmers_counts_reference_kats(ctx, golden_dir):
    # tests/parser_unittest.h:38-63 on tests/testdata/test.fa
    path = os.path.join(golden_dir, "test.fa")
    for k, complements, case_sensitive, want in [(10, True, False, 3), (10, True, True, 2), (5, True, False, 11), (5, False, False, 11),
                                                  (2, True, False, 9), (2, False, False, 11)]:
        s = kc.ReadKMers(ctx, path, k, complements, case_sensitive)
        assert s.size() == want
        s.close()


def test_empty_and_short_inputs(ctx):
    s = kc.KmerSet(ctx, 31, True)
    s.AddKMers("")
    s.AddKMers("ACGT")
    s.AddKMers("N" * 100)
    assert s.size() == 0
    with pytest.raises(kc.KcError) as e:
        kc.get_simplitigs(s)
    assert e.value.code == kc.api.KC_E_EMPTY
    s.close()
    with pytest.raises(kc.KcError) as e:
        kc.compute(ctx, b"ACGTNNNN\nAC", 5)
    assert e.value.code == kc.api.KC_E_NO_KMERS
    with pytest.raises(kc.KcError) as e:
        kc.simplitigs_from_strings(ctx, [], 3)
    assert e.value.code == kc.api.KC_E_EMPTY
    with pytest.raises(kc.KcError) as e:
        kc.simplitigs_from_strings(ctx, ["ACGN"], 3)   # assert in simplitigs.h:82
    assert e.value.code == kc.api.KC_E_INVALID_CHAR
    with pytest.raises(kc.KcError):
        kc.KmerSet(ctx, 128, True)


# ------------------------------------------------------------------------------------------------ B: simplitigs (T-C)
@pytest.mark.parametrize("k", [3, 4, 8, 13, 31, 32, 63, 64, 127])
@pytest.mark.parametrize("complements", [True, False])
def test_simplitigs_match_canonical_oracle(ctx, oracle, k, complements):
    rng = random.Random(77 + k)
    cases = []
    cases.append("\n".join(r.decode() for r in gen.stress_reads(4000, 120, 150 if k < 100 else 300, seed=k)))
    cases.append(_rand_seq(rng, 3000, "AC"))            # low complexity: heavy branching, palindromes for even k, cycles
    cases.append("A" * 200 + "C" * 50 + "ACACACACACACACACAC" * 10)
    cases.append(_rand_seq(rng, 2000) * 2)              # repeat
    circ = _rand_seq(rng, 400)
    cases.append(circ + circ[:k - 1] + circ[: k])      # closes a cycle in the de Bruijn graph
    big = _rand_seq(rng, 6000)
    cases.append(big + big[:k - 1] + "N" + _rand_seq(rng, 500))   # a cycle of 6000 k-mers (beyond the parallel cutter's 4096 steps) + a chain
    cases.append(big + big[:k - 1])                               # nothing but that cycle: no chain end anywhere
    cases.append(circ + circ[:k - 1])                             # nothing but a small cycle
    for seq in cases:
        want = oracle.simplitigs_of_sequence(seq.replace("\n", "N"), k, complements, mode=1)
        s = kc.KmerSet(ctx, k, complements, capacity_hint=len(seq))
        s.AddKMers(seq)
        sim = kc.get_simplitigs(s)
        got = sim.strings()
        assert got == want
        # every k-mer of the set is covered exactly once
        n, total = sim.count()
        assert total - n * (k - 1) == s.size()
        sim.close(); s.close()


@pytest.mark.parametrize("k,links", [(4, "default"), (13, "default"), (31, "default"), (40, "default"), (127, "default"),
                                     (31, "plain"), (127, "plain"), (31, "plain-dense-marks"), (63, "default")])
def test_segment_parallel_walks_forced(ctx, oracle, k, links, monkeypatch):
    # KC_CHAIN_LIMIT=2 makes every chain of more than 2 k-mers "long": the splitter/segment path handles all of them.  On minimizer
    # tables (k >= 18) the segments hop over contracted runs by default; KC_SEG_PLAIN keeps the plain links (one step per k-mer),
    # KC_SEG_DENSE_MARKS the per-node marks of the plain walks.
    monkeypatch.setenv("KC_CHAIN_LIMIT", "2")
    if links != "default":
        monkeypatch.setenv("KC_SEG_PLAIN", "1")
    if links == "plain-dense-marks":
        monkeypatch.setenv("KC_SEG_DENSE_MARKS", "1")
    rng = random.Random(31 + k)
    cases = ["\n".join(r.decode() for r in gen.stress_reads(6000, 150, 200 if k < 100 else 400, seed=100 + k)),
             _rand_seq(rng, 30000), _rand_seq(rng, 3000, "AC"), _rand_seq(rng, 1500) * 3]
    for complements in (True, False):
        for seq in cases:
            want = oracle.simplitigs_of_sequence(seq.replace("\n", "N"), k, complements, mode=1)
            s = kc.KmerSet(ctx, k, complements, capacity_hint=len(seq))
            s.AddKMers(seq)
            sim = kc.get_simplitigs(s)
            assert sim.strings() == want
            # downstream phases read firstK/lastK written by the segment emitter
            rc, wout, wmask = oracle.global_(want, k, complements, want_mask=True)
            out, mask, _ = kc.Global(sim, complements, want_mask=True, kmer_set=s)
            assert (out, mask) == (wout, wmask)
            sim.close(); s.close()


@pytest.mark.parametrize("knobs", [{"KC_BUCKET_MIN": "0"}, {"KC_BUCKET_MIN": "1000000000"}, {"KC_NOREC": "1"}, {"KC_NOSTAGE": "1"},
                                   {"KC_BUCKET_MIN": "0", "KC_LOCAL_PROBES": "2"}, {"KC_NOFRAG": "1"}, {"KC_CHAIN_LIMIT": "600"},
                                   {"KC_BUCKET_MIN": "0", "KC_SCATTER_ROLL": "1"}, {"KC_BUCKET_MIN": "0", "KC_PROBE_LISTED_GLOBAL": "1"},
                                   {"KC_BUCKET_MIN": "0", "KC_BUILD_NOPREFETCH": "1", "KC_SCATTER_CHUNKS": "2"},
                                   {"KC_BUCKET_MIN": "0", "KC_NO_BUILD_IDS": "1"}, {"KC_BUCKET_MIN": "0", "KC_SCATTER_V1": "1"},
                                   {"KC_WALK_BOTH": "1"}, {"KC_WALK_CLASSES_MIN": "0"}],
                         ids=["bucketed-build", "direct-insert", "no-recording", "no-tag-staging", "bucketed-with-spills", "plain-walk",
                              "fragment-contraction-then-restore", "rolling-scatter", "listed-ports-from-the-table", "no-build-prefetch",
                              "node-ids-by-a-sweep-of-their-own", "wave-scatter-of-round-4", "every-head-walks", "walk-classes-on-a-small-input"])
def test_alternative_device_paths_give_identical_results(ctx, oracle, knobs, monkeypatch):
    """Every path the library picks by input size (bucketed LDS build vs direct CAS inserts, recorded-stream emit vs chain
    walk, LDS tag staging; the wave-cooperative bucket scatter -- default with KC_BUCKET_MIN=0 for single-word keys on minimizer
    tables -- against the rolling one) must give the same set and the same simplitigs; KC_LOCAL_PROBES=2 makes block-local probe
    runs overflow, so the spill list + global second home of the bucketed build are exercised too.  Round 5: the scatter of round 4
    against the leaner one (the default here), every head walking its chain against one walk per chain with turned recordings (the
    default).  (Buckets that run full and send keys to the scatter's side list: the low-complexity and repeat cases of this file -- 900
    to 1 500 keys aside each, and 10^6 where the list runs full and the second sweep takes over; `KC_SET_TRACE=1` shows them.)"""
    for name, value in knobs.items():
        monkeypatch.setenv(name, value)
    rng = random.Random(4242)
    for k in (16, 18, 23, 31):      # single-word keys: plain hashing (16) and minimizer blocks (k >= 18)
        reads = [r.decode() for r in gen.stress_reads(20000, 200, 250, seed=500 + k)]
        seq = "\n".join(reads)
        for complements in (True, False):
            want_set = sorted(oracle.kmer_set(seq.replace("\n", "N"), k, complements))
            # one call, and the same records added in three calls (the later ones update a non-empty table)
            s1 = kc.KmerSet(ctx, k, complements, capacity_hint=len(seq))
            s1.AddKMers(seq)
            s2 = kc.KmerSet(ctx, k, complements)
            third = len(reads) // 3
            for part in (reads[:third], reads[third:2 * third], reads[2 * third:] + reads[:5]):
                s2.AddKMers("\n".join(part))
            assert s1.size() == len(want_set) and s2.size() == len(want_set)
            assert sorted(s1.keys()) == want_set and sorted(s2.keys()) == want_set
            probe = want_set[:200] + [(x + 1) & ((1 << (2 * k)) - 1) for x in want_set[:200]]
            wset = set(want_set)
            assert s2.contains(probe) == [x in wset for x in probe]
            want = oracle.simplitigs_of_sequence(seq.replace("\n", "N"), k, complements, mode=1)
            for s in (s1, s2):
                sim = kc.get_simplitigs(s)
                assert sim.strings() == want
                sim.close()
            s1.close(); s2.close()


@pytest.mark.parametrize("roll", [False, True], ids=["wave-scatter", "rolling-scatter"])
def test_bucket_scatter_edge_cases(ctx, oracle, roll, monkeypatch):
    """The two bucket scatter kernels (kc_bucket_scatter_wave_kernel: a lane is a position; kc_bucket_scatter_kernel: a thread rolls
    a stretch) on what their position arithmetic has to get right: records shorter than k and of exactly k, N and lower case
    anywhere (case-sensitive inserts: the first base decides, parser.h:41-42), lengths around the 64-position chunks and the 16-chunk
    stretches of a wave, single-word and wide keys with a minimizer table."""
    monkeypatch.setenv("KC_BUCKET_MIN", "0")
    if roll:
        monkeypatch.setenv("KC_SCATTER_ROLL", "1")
    rng = random.Random(99)
    for k in (18, 19, 24, 31, 32, 33, 47, 63, 64, 65, 100, 127):   # (wide keys: kc_bucket_scatter_wave_wide_kernel, windows of up to 110 positions)
        for total in (k, 63, 64, 65, 1023, 1024, 1025, 1024 + k, 5000, 70000):
            recs, n = [], 0
            while n < total:
                ln = rng.choice([1, k - 1, k, k + 1, 2 * k, 100, 333]) if rng.random() < 0.7 else rng.randint(1, 2000)
                ln = min(ln, total - n)
                alphabet = rng.choice(["ACGT", "ACGTN", "ACGTacgt", "ACGTacgtN"])
                recs.append("".join(rng.choice(alphabet) for _ in range(ln)))
                n += ln + 1
            seq = "\n".join(recs)
            for complements in (True, False):
                for cs in (False, True):
                    want = sorted(oracle.kmer_set(seq.replace("\n", "N"), k, complements, case_sensitive=cs))
                    s1 = kc.KmerSet(ctx, k, complements, capacity_hint=len(seq))
                    s1.AddKMers(seq, case_sensitive=cs)
                    assert sorted(s1.keys()) == want, (k, total, complements, cs)
                    s1.close()


@pytest.mark.parametrize("k", [21, 31, 63, 127])
def test_fragment_contraction_path(ctx, k, monkeypatch):
    """Read-like input takes the fragment path by default: in-block chain runs are contracted into shortcuts in LDS and the
    measure walk hops from fragment to fragment (kc_frag_contract_kernel).  Same bytes as the plain walk (KC_NOFRAG=1) on
    chains of ~500 nodes (every fragment length incl. runs beyond the 19 inline bases), both strand modes; with a chain limit
    below some chain's length, and with records whose k-mers form cycles, the hop walk is abandoned, the logged link words
    are put back (kc_frag_restore_kernel) and the classic flow (segment-parallel walk / cycle cut) takes over."""
    rng = np.random.default_rng(77 + k)
    unit = rng.integers(0, 4, size=3 * k + 40)
    cyc = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate([unit, unit, unit[:k]])]   # the k-mers of this record form a cycle
    reads = gen.random_joined(3000, 500 + k, seed=900 + k)
    long_reads = gen.random_joined(60, 2500 + k, seed=901 + k)   # a few chains beyond the limit of the second setting
    nl = np.array([10], dtype=np.uint8)
    inputs = {"reads": np.concatenate([reads, nl, long_reads]), "reads+cycles": np.concatenate([reads, nl, cyc, nl, cyc[::-1].copy()])}
    for name, seq in inputs.items():
        for complements in (True, False):
            monkeypatch.setenv("KC_NOFRAG", "1")
            want = kc.compute(ctx, seq, k, complements)
            w = want.array().copy()
            cycles = want.stats["cycles_cut"]
            want.close()
            monkeypatch.delenv("KC_NOFRAG")
            assert (cycles > 0) == (name == "reads+cycles")
            for env in ({}, {"KC_CHAIN_LIMIT": "1200"}, {"KC_CHAIN_LIMIT": "1200", "KC_SEG_PLAIN": "1"}, {"KC_CHAIN_LIMIT": "40", "KC_BSHIFT": "10"}, {"KC_BSHIFT": "10"},
                        {"KC_EMIT_LANES": "64"}, {"KC_EMIT_LANES": "8"},    # (phase 3: a wavefront / 8 lanes per vertex)
                        {"KC_WALK_CLASSES_MIN": "0"}, {"KC_WALK_CLASSES_MIN": "0", "KC_BSHIFT": "10"}):   # (the class passes of the walks, which inputs this small do not take by themselves)
                for var, value in env.items():
                    monkeypatch.setenv(var, value)
                got = kc.compute(ctx, seq, k, complements)
                assert np.array_equal(got.array(), w), (k, name, complements, env)
                assert got.stats["cycles_cut"] == cycles
                got.close()
                for var in env:
                    monkeypatch.delenv(var)


@pytest.mark.parametrize("k", [21, 45])
def test_overlap_hit_lists_bucketed_sort_and_its_fallback(ctx, oracle, k, monkeypatch):
    """Hit lists beyond 1024 rows are ordered by an MSD bucket pass + one-workgroup LDS sorts (kcu_sort_rows_bucketed); keys
    that share their leading characters (here: unidirectional runs over a two-letter alphabet, so the top key bits take few
    values) overflow a bucket and the level falls back to the LSD radix sort.  Both routes, and the LSD sort alone
    (KC_NO_BUCKET_SORT=1), must give the same bytes; the smaller case is also checked against the oracle."""
    rng = np.random.default_rng(500 + k)
    def reads(n, length, alphabet):
        a = np.frombuffer(alphabet.encode(), dtype=np.uint8)[rng.integers(0, len(alphabet), size=(n, length + 1))]
        a[:, length] = 10
        return a.reshape(-1)[:-1].copy()
    cases = {"skewed": reads(40000, k + 25, "AC"), "even": reads(30000, k + 40, "ACGT")}
    for name, seq in cases.items():
        for complements in (False, True):
            got = kc.compute(ctx, seq, k, complements)
            g = got.array().copy(); got.close()
            monkeypatch.setenv("KC_NO_BUCKET_SORT", "1")
            ref = kc.compute(ctx, seq, k, complements)
            r = ref.array().copy(); ref.close()
            monkeypatch.delenv("KC_NO_BUCKET_SORT")
            assert np.array_equal(g, r), (name, k, complements)
    small = reads(3000, k + 25, "AC")
    text = small.tobytes().decode()
    for complements in (False, True):
        want = oracle.simplitigs_of_sequence(text.replace("\n", "N"), k, complements, mode=1)
        rc, wout, wmask = oracle.global_(want, k, complements, want_mask=True)
        got = kc.compute(ctx, small, k, complements)
        assert rc == 0 and got.array().tobytes().decode() == wout
        got.close()


@pytest.mark.parametrize("k", [23, 31, 63])
def test_high_levels_through_the_16mer_table(ctx, k, monkeypatch):
    """Levels d >= 16 of phase 2 run through one table of the prefix ends' first 16-mers from 2^16 vertices on
    (KC_T16_MIN=0 forces it, a huge value turns it off): same bytes on read sets with overlapping reads (many hits at high
    levels), and on ends that share their first 16 bases thousands of times (chains beyond the limit: those levels fall
    back to a table of their own)."""
    rng = np.random.default_rng(900 + k)
    stem = rng.integers(0, 4, size=40)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rep = []
    for _ in range(400):   # 400 records that all start with the same 40 bases
        rep.append(acgt[np.concatenate([stem, rng.integers(0, 4, size=k + 20)])])
        rep.append(np.array([10], dtype=np.uint8))
    repetitive = np.concatenate(rep)[:-1]
    reads = np.frombuffer("\n".join(r.decode() for r in gen.stress_reads(30000, 900, max(150, 2 * k), seed=k, n_frac=0.0, lower_frac=0.0)).encode(), dtype=np.uint8)
    for seq in (reads, repetitive):
        for complements in (True, False):
            monkeypatch.setenv("KC_T16_MIN", "1000000000000")
            a = kc.compute(ctx, seq, k, complements); want = a.array().copy(); a.close()
            monkeypatch.setenv("KC_T16_MIN", "0")
            monkeypatch.setenv("KC_T16_FORCE", "1")   # (also where the table would be judged not to pay: many ends per 16-mer)
            b = kc.compute(ctx, seq, k, complements); got = b.array().copy(); b.close()
            monkeypatch.delenv("KC_T16_MIN"); monkeypatch.delenv("KC_T16_FORCE")
            assert np.array_equal(got, want), (k, complements, len(seq))


def test_sampled_path_ranking_equals_pointer_jumping(ctx, monkeypatch):
    """Phase 3 ranks the path by pointer jumping over all vertices (small inputs) or over sampled splitters with one walk per
    segment (from 2^20 vertices on; KC_SAMPLED_RANK_MIN=0 forces it): both must place every block at the same offset --
    read sets (thousands of short blocks), a few long records (one-vertex paths among them) and both strand modes."""
    cases = [gen.random_joined(4000, 300, seed=31), gen.random_joined(3, 20000, seed=32),
             np.concatenate([gen.random_joined(500, 90, seed=33), np.array([10], dtype=np.uint8), gen.random_joined(1, 50000, seed=34)])]
    for seq in cases:
        for k in (31, 63):
            for complements in (True, False):
                a = kc.compute(ctx, seq, k, complements, want_mask=True)
                want = a.strings(want_mask=True); a.close()
                monkeypatch.setenv("KC_SAMPLED_RANK_MIN", "0")
                b = kc.compute(ctx, seq, k, complements, want_mask=True)
                got = b.strings(want_mask=True); b.close()
                monkeypatch.delenv("KC_SAMPLED_RANK_MIN")
                assert got == want, (len(seq), k, complements)


def test_result_write_mem_equals_copy(ctx):
    """kc_result_write_mem (pinned chunks moved on by several threads; what the CLI fills its output mapping with) gives the
    bytes of kc_result_copy, for the superstring and for the -M mask, at a size beyond one 32 MB chunk."""
    import ctypes
    seq = gen.random_joined(60000, 700, seed=77)
    res = kc.compute(ctx, seq, 31, True, want_mask=True)
    want = res.strings(want_mask=True)
    assert res.length > (32 << 20)
    for which in (0, 1):
        buf = np.zeros(res.length, dtype=np.uint8)
        ctx.check(ctx.lib.kc_result_write_mem(res.h, buf.ctypes.data_as(ctypes.c_void_p), which))
        assert buf.tobytes().decode() == want[which]
    res.close()


@pytest.mark.parametrize("k", [31, 40, 100])
def test_bucket_overflow_falls_back_to_direct_inserts(ctx, oracle, k, monkeypatch):
    """A short read repeated thousands of times sends more occurrences to a few table blocks than a bucket holds: the
    bucketed build keeps what fits, raises its overflow flag, and the direct insert kernel adds the rest (1 / 2 / 4 key words)."""
    monkeypatch.setenv("KC_BUCKET_MIN", "0")
    rng = random.Random(900 + k)
    read = _rand_seq(rng, k + 300)
    other = [_rand_seq(rng, 400) for _ in range(50)]
    seq = "\n".join([read] * 4000 + other)
    want = sorted(set(oracle.kmer_set(read, k, True)) | set(x for o in other for x in oracle.kmer_set(o, k, True)))
    s = kc.KmerSet(ctx, k, True, capacity_hint=len(seq))
    s.AddKMers(seq)
    assert s.size() == len(want) and sorted(s.keys()) == want
    sim = kc.get_simplitigs(s)
    assert sim.strings() == oracle.simplitigs_of_sequence(seq.replace("\n", "N"), k, True, mode=1)
    sim.close(); s.close()


def test_randomised_parity_run():
    """20 s of tools/fuzz_parity.py: random shapes (reads, low complexity, repeats, circles, thousands of copies of one read),
    random k in 2..127, both strand modes, both set-build paths, forced segment walks -- whole `compute` output vs the oracle."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "20", "7"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "fuzz ok" in p.stdout, p.stdout + p.stderr


def test_a_launch_beyond_the_grid_limit_fails_the_call_with_e_arg():
    """The launch guard through a real call: with KC_TEST_GRID_LIMIT the library treats launches of that many threads as it
    treats 2^32 -- the call comes back with KC_E_ARG and the '2^32' message, whatever launch trips first, instead of running on with the
    output of a skipped kernel; small inputs in the same process stay unaffected (the flag does not go stale)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import kmercamel_amd as kc, gen\n"
        "c = kc.Context(0)\n"
        "small = b'\\n'.join(gen.random_records(2, 50, seed=1))\n"
        "big = b'\\n'.join(gen.random_records(3000, 400, seed=2))\n"
        "a = kc.compute(c, small, 31, True).strings()[0]\n"
        "for want_mask in (False, True):\n"
        "    try:\n"
        "        kc.compute(c, big, 31, True, want_mask=want_mask); print('NO ERROR')\n"
        "    except kc.KcError as e:\n"
        "        print('CODE', e.code, '2^32' in str(e))\n"
        "    assert kc.compute(c, small, 31, True).strings()[0] == a\n"
        "print('done')\n" % (root, os.path.join(root, "tests")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, KC_TEST_GRID_LIMIT="4000"))
    assert p.returncode == 0 and p.stdout.count("CODE %d True" % kc.api.KC_E_ARG) == 2 and "done" in p.stdout, p.stdout + p.stderr


def _export_keys_np(s):
    """the set's keys as a numpy array (single-word keys), through the C ABI"""
    import ctypes
    import numpy as np
    n = s.size()
    buf = np.empty(max(1, n), dtype=np.uint64)
    s.ctx.check(s.ctx.lib.kc_set_export(s.h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    return buf[:n]


def test_full_size_properties_cfg1(ctx):
    """BASELINE configs[1] at full size (100 000 records x 1 030 bp, k = 31: 10^8 k-mers), checked through properties that
    do not need the oracle: (a) two runs give identical bytes (the result does not depend on the schedule); (b) min-one
    mask: ones == N; (c) round trip: the k-mers starting at the upper-case positions of the output (re-read with the
    case-sensitive AddKMers, as maskopt does) are exactly the input's k-mer set (equal size, equal order-free checksums,
    spot checks both ways); (d) length bounds N + k - 1 <= L <= N + S(k-1); (e) -M has at least as many ones and the
    same characters."""
    import hashlib
    import numpy as np
    k = 31
    seq = gen.random_joined(100000, 1030, seed=1)
    raw = seq.tobytes()
    r = kc.compute(ctx, raw, k, True, want_mask=True)
    N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
    out, mask = r.strings(want_mask=True)
    r.close()
    assert 100000 * 1000 - 20 <= N <= 100000 * 1000          # i.i.d. sequence: duplicates are (almost) impossible
    assert N + k - 1 <= L <= N + S * (k - 1)
    o = np.frombuffer(out.encode(), dtype=np.uint8)
    m = np.frombuffer(mask.encode(), dtype=np.uint8)
    assert int((o < 97).sum()) == N                              # (b)
    assert int((m < 97).sum()) >= N and np.array_equal(o & 0xDF, m & 0xDF)   # (e)
    r2 = kc.compute(ctx, raw, k, True, want_mask=False)        # (a)
    out2, _ = r2.strings()
    r2.close()
    assert hashlib.md5(out2.encode()).hexdigest() == hashlib.md5(out.encode()).hexdigest()
    # (c) round trip through the sets
    a = kc.KmerSet(ctx, k, True, capacity_hint=len(raw))
    a.AddKMers(raw)
    b = kc.KmerSet(ctx, k, True, capacity_hint=len(out))
    b.AddKMers(out, case_sensitive=True)
    assert a.size() == N and b.size() == N
    ka, kb = _export_keys_np(a), _export_keys_np(b)
    mix = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        assert int(ka.sum(dtype=np.uint64)) == int(kb.sum(dtype=np.uint64))
        assert int(np.bitwise_xor.reduce(ka * mix)) == int(np.bitwise_xor.reduce(kb * mix))
    probe = [int(x) for x in ka[:: max(1, N // 500)][:500]]
    assert all(b.contains(probe)) and all(a.contains([int(x) for x in kb[:: max(1, N // 500)][:500]]))
    a.close(); b.close()


def _export_key_words_np(s, W):
    """the set's keys as an (n, W) numpy array of 64-bit words (kc_set_export: W words per key), through the C ABI"""
    import ctypes
    import numpy as np
    n = s.size()
    buf = np.empty(max(1, n * W), dtype=np.uint64)
    s.ctx.check(s.ctx.lib.kc_set_export(s.h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    return buf[: n * W].reshape(n, W)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [63, 127])
def test_full_size_properties_wide_keys(ctx, k):
    """BASELINE configs[4] at full size: the wide-key paths (k = 63: two key words, k = 127: four) on 100 000 synthetic records of
    k - 1 + 1000 bp (10^8 k-mers), through the same oracle-free properties as test_full_size_properties_cfg1: two runs give identical
    bytes, ones == N, the length bounds, -M has at least as many ones over the same characters, and the k-mer set read back off the
    output equals the input's set (sizes, order-free checksums over every key word, spot checks both ways)."""
    import hashlib
    import numpy as np
    W = 2 if k < 64 else 4
    seq = gen.random_joined(100000, k - 1 + 1000, seed=1)
    raw = seq.tobytes()
    r = kc.compute(ctx, raw, k, True, want_mask=True)
    N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
    out, mask = r.strings(want_mask=True)
    r.close()
    assert N == 100000 * 1000                                   # i.i.d. sequence: no duplicate among 10^8 k-mers of 63 / 127 bases
    assert N + k - 1 <= L <= N + S * (k - 1)
    o = np.frombuffer(out.encode(), dtype=np.uint8)
    m = np.frombuffer(mask.encode(), dtype=np.uint8)
    assert int((o < 97).sum()) == N
    assert int((m < 97).sum()) >= N and np.array_equal(o & 0xDF, m & 0xDF)
    r2 = kc.compute(ctx, raw, k, True, want_mask=False)
    out2, _ = r2.strings()
    r2.close()
    assert hashlib.md5(out2.encode()).hexdigest() == hashlib.md5(out.encode()).hexdigest()
    a = kc.KmerSet(ctx, k, True, capacity_hint=len(raw))
    a.AddKMers(raw)
    b = kc.KmerSet(ctx, k, True, capacity_hint=len(out))
    b.AddKMers(out, case_sensitive=True)
    assert a.size() == N and b.size() == N
    ka, kb = _export_key_words_np(a, W), _export_key_words_np(b, W)
    mix = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        for w in range(W):
            assert int(ka[:, w].sum(dtype=np.uint64)) == int(kb[:, w].sum(dtype=np.uint64))
        # a checksum that ties the words of one key together (a per-word sum would not notice words swapped between keys)
        ha = ka[:, 0] * mix
        hb = kb[:, 0] * mix
        for w in range(1, W):
            ha = (ha ^ ka[:, w]) * mix
            hb = (hb ^ kb[:, w]) * mix
        assert int(np.bitwise_xor.reduce(ha)) == int(np.bitwise_xor.reduce(hb)) and int(ha.sum(dtype=np.uint64)) == int(hb.sum(dtype=np.uint64))

    def as_int(row):
        v = 0
        for w in range(W):
            v |= int(row[w]) << (64 * w)
        return v
    step = max(1, N // 500)
    assert all(b.contains([as_int(x) for x in ka[::step][:500]])) and all(a.contains([as_int(x) for x in kb[::step][:500]]))
    a.close(); b.close()


def test_coverage_reads_with_errors_at_scale(ctx, monkeypatch):
    """The shape real sequencing data has and the benchmark's i.i.d. reads do not: 10x coverage of a 5 Mbp genome by 150 bp
    reads from both strands with 1 % substitution errors -- every k-mer many times over, millions of branch ports, overlaps
    at every level from k-1 down, cycles.  ~2 * 10^7 distinct 31-mers: too many for the oracle, so (a) the oracle-free
    properties (ones == N, length bounds, round trip of the k-mer set) and (b) byte-identity of the default device paths
    with their alternatives (plain walk, per-level tables only, LSD sorts only, single-cursor edge log, plain path ranking)."""
    rng = np.random.default_rng(20260)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    G, n_reads, rl, k = 5_000_000, 330_000, 150, 31
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    pos = rng.integers(0, G - rl, size=n_reads)
    codes = genome[pos[:, None] + np.arange(rl)[None, :]]
    rev = rng.random(n_reads) < 0.5
    codes[rev] = 3 - codes[rev][:, ::-1]
    err = rng.random(codes.shape) < 0.01
    codes[err] = (codes[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
    a = np.empty((n_reads, rl + 1), dtype=np.uint8)
    a[:, :rl] = acgt[codes]
    a[:, rl] = 10
    seq = a.reshape(-1)[:-1].copy()
    r = kc.compute(ctx, seq, k, True)
    N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
    o = r.array().copy()
    rounds = r.stats["matching_rounds"] if "matching_rounds" in r.stats else None
    r.close()
    assert 10_000_000 < N < 40_000_000 and S > 100_000          # genome k-mers + ~30 per error
    assert N + k - 1 <= L <= N + S * (k - 1)
    assert int(np.count_nonzero(o < 97)) == N
    sa = kc.KmerSet(ctx, k, True, capacity_hint=len(seq)); sa.AddKMers(seq)
    sb = kc.KmerSet(ctx, k, True, capacity_hint=len(o)); sb.AddKMers(o, case_sensitive=True)
    assert sa.size() == N and sb.size() == N
    ka, kb = _export_keys_np(sa), _export_keys_np(sb)
    mix = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        assert int(ka.sum(dtype=np.uint64)) == int(kb.sum(dtype=np.uint64))
        assert int(np.bitwise_xor.reduce(ka * mix)) == int(np.bitwise_xor.reduce(kb * mix))
    sa.close(); sb.close()
    for env in ({"KC_NOFRAG": "1"}, {"KC_T16_MIN": "1000000000000"}, {"KC_T16_FORCE": "1"}, {"KC_NO_BUCKET_SORT": "1"}, {"KC_NO_SHARDED_LOG": "1"},
                {"KC_SAMPLED_RANK_MIN": "1000000000000"}, {"KC_SAMPLED_RANK_MIN": "0"},
                # phase 2: the host-driven loop for every level / the grouped levels for every table level / no wavefront replays in
                # them (every larger bucket sends its level back to the host-driven loop) / eager compaction, no level windows
                {"KC_OV_HOST_LOOP": "1"}, {"KC_OVG_MAX": "1000000000000"}, {"KC_OVG_MAX": "1000000000000", "KC_OVG_NO_LDS": "1"},
                {"KC_OV_EAGER_SHRINK": "1", "KC_OV_NO_WINDOW": "1"},
                # small buckets replayed from global memory by one thread each (not out of a lane's LDS), in both loops
                {"KC_OV_NO_LANE_REPLAY": "1"}, {"KC_OV_NO_LANE_REPLAY": "1", "KC_OVG_MAX": "1000000000000"},
                # phase 3: 8 lanes per vertex (the form for short simplitigs) on these 1 030-character ones
                {"KC_EMIT_LANES": "8"}, {"KC_EMIT_LANES": "64"},
                # phase 1, simplitigs out of the recordings: one / two lanes per simplitig (the forms for short ones) on these 33-word ones
                {"KC_STREAM_LANES": "1"}, {"KC_STREAM_LANES": "2"}):
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        r2 = kc.compute(ctx, seq, k, True)
        same = np.array_equal(r2.array(), o)
        r2.close()
        for name in env:
            monkeypatch.delenv(name)
        assert same, env


def test_full_size_properties_cfg2(ctx):
    """BASELINE configs[2] at full size (2 500 000 records x 1 030 bp, k = 31: 2.5 * 10^9 k-mers, the set bench.py runs),
    through the same oracle-free properties as cfg1: determinism (two runs, identical bytes), ones == N (min-one mask),
    length bounds, and the round trip -- the k-mers at the upper-case positions of the output are exactly the input's set
    (equal sizes, equal order-free checksums over all 2.5 G keys, spot checks both ways)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    k = 31
    seq = bench.generate(2, 0, 2500000, 1030)
    r = kc.compute(ctx, seq, k, True, want_mask=False)
    N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
    o = r.array()
    r.close()
    assert 2500000 * 1000 - 200 <= N <= 2500000 * 1000
    assert N + k - 1 <= L <= N + S * (k - 1)
    assert int(np.count_nonzero(o < 97)) == N
    st1 = dict(r.stats)
    r2 = kc.compute(ctx, seq, k, True, want_mask=False)
    o2 = r2.array()
    st2 = dict(r2.stats)
    r2.close()
    assert np.array_equal(o, o2)
    del o2
    # regression guard on the set build (round 5 review, item 5): variants of the bucket scatter that LOOKED cheaper ran 28-36 ms
    # instead of 17-18 ms depending on what the device had done before ("the scatter's time does not follow its instruction
    # count"; never explained) -- the build's two kernels together take 39-41 ms at this size, a scatter that has fallen into the slow
    # mode 49 ms or more.  The better of the two passes is held against 46 ms (boxes differ by ~7 %).
    build_ms = min(st1["us_insert_kernel"], st2["us_insert_kernel"]) / 1000.0
    assert build_ms <= 46.0, "set build (scatter + block build) took %.1f ms at 2.5e9 k-mers: the scatter has regressed" % build_ms
    mix = np.uint64(0x9E3779B97F4A7C15)

    def digest(s):
        keys = _export_keys_np(s)
        with np.errstate(over="ignore"):
            d = (len(keys), int(keys.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(keys * mix)))
        return d, [int(x) for x in keys[:: max(1, len(keys) // 500)][:500]]
    a = kc.KmerSet(ctx, k, True, capacity_hint=len(seq))
    a.AddKMers(seq)
    da, pa = digest(a)
    a.close()
    b = kc.KmerSet(ctx, k, True, capacity_hint=L)
    b.AddKMers(o, case_sensitive=True)
    db, pb = digest(b)
    assert da[0] == N and da == db
    assert all(b.contains(pa))
    b.close()
    a = kc.KmerSet(ctx, k, True, capacity_hint=len(seq))
    a.AddKMers(seq)
    assert all(a.contains(pb))
    a.close()


def test_full_size_beyond_2pow32_positions(ctx):
    """4.4 * 10^9 k-mers (4 400 000 records x 1 030 bp = 4.53 * 10^9 input positions, more than 2^32; ~215 GB of the 288 GB): every kernel
    that sweeps positions or table slots runs on 2-D grids here, the table is built at load 0.66 instead of 0.5 (the memory rule of
    kc_set_create), and ~1.4 * 10^8 ports go to the generic lookups -- whose list once overflowed at this size, sending all of round 1 through
    the generic kernel (1.9 s per pass instead of 0.45 s: round 5).  Oracle-free properties as for cfg2, plus a bound on the phase that fell
    off the cliff."""
    import sys
    import numpy as np
    import ctypes
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip = ctypes.CDLL("libamdhip64.so")   # (the context of the suite has the device open: this only asks for its size)
    if hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) != 0 or total_b.value < 250 * (1 << 30):
        pytest.skip("needs a 288 GB device")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    k, R = 31, 4400000
    seq = bench.generate(2, 0, R, 1030)
    assert len(seq) > (1 << 32)
    r = kc.compute(ctx, seq, k, True, want_mask=False)
    N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
    o = r.array()
    r.close()
    assert R * 1000 - 200 <= N <= R * 1000 and R <= S <= R + 200
    assert N + k - 1 <= L <= N + S * (k - 1)
    assert int(np.count_nonzero(o < 97)) == N
    b = kc.KmerSet(ctx, k, True, capacity_hint=L)
    b.AddKMers(o, case_sensitive=True)
    assert b.size() == N
    rng = np.random.default_rng(3)
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    qs = []
    for p0 in rng.integers(0, len(seq) - k, size=1500):   # sampled input k-mers (also from beyond position 2^32) are represented
        w = code[seq[p0:p0 + k]]
        if (w == 255).any():
            continue
        f = 0; rc = 0
        for j, c in enumerate(w.tolist()):
            f = (f << 2) | c
            rc |= (3 - c) << (2 * j)
        qs.append(min(f, rc))
    assert len(qs) > 1000 and all(b.contains(qs))
    b.close()
    del o
    r2 = kc.compute(ctx, seq, k, True, want_mask=False)   # (pool warm: the phase times are those of a steady pass)
    t1 = r2.stats["us_simplitig_phase"]
    r2.close()
    assert t1 < 900000, "phase 1 took %.0f ms at 4.4 G k-mers (0.29 s when the spill list holds)" % (t1 / 1e3)


def test_simplitigs_reference_kats(ctx):
    # tests/simplitigs_unittest.h:183-208 (order-tolerant there: sorted by length)
    enc = oracle_lib.encode_kmer
    for kmers, k, want in [(["GCT", "TAA", "AAA"], 3, ["GCT", "TAAA"]), (["TAA", "AAA", "GCT"], 3, ["GCT", "TAAA"]),
                           (["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTG", "TTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"], 31, ["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"])]:
        s = kc.KmerSet(ctx, k, False)
        s.AddKMers("N".join(kmers))
        assert sorted(s.keys()) == sorted(enc(x) for x in kmers)
        got = kc.get_simplitigs(s).strings()
        assert sorted(got, key=len) == want
        s.close()


# ------------------------------------------------------------------------------------------------ C1: overlap path (T-A)
OHP_KATS = [  # tests/global_unittest.h:53-111
    (["AT"], 2, True, False, [NONE, NONE], [255, 255]),
    (["ACG", "TAC", "GGC"], 3, False, False, [2, 0, NONE], [1, 2, 255]),
    (["ACAA", "ATTT", "AACA"], 4, True, False, [4, 3, 0, 5, NONE, NONE], [2, 2, 3, 3, 255, 255]),
    (["ACG", "CGT", "TAA"], 3, False, True, [1, 2, 0], [2, 1, 1]),
    (["ACGT", "TAA"], 3, False, True, [1, 0], [1, 1]),
    (["ACC", "CGG"], 3, True, True, [3, 2, 0, 1], [2, 2, 0, 2]),
]


@pytest.mark.parametrize("strs,k,complements,lb,want_ef,want_ov", OHP_KATS)
def test_overlap_path_reference_kats(ctx, strs, k, complements, lb, want_ef, want_ov):
    s = kc.simplitigs_from_strings(ctx, strs, k)
    ef, ov = kc.OverlapHamiltonianPath(s, complements, lb)
    assert ef == want_ef and ov == want_ov
    s.close()


def _random_simplitig_set(rng, k, n, alphabet):
    base = _rand_seq(rng, max(k, 600), alphabet)
    out = []
    for _ in range(n):
        L = k + rng.choice([0, 0, 1, 2, 5, 20, 100])
        if rng.random() < 0.6 and len(base) >= L:
            p = rng.randrange(0, len(base) - L + 1)
            out.append(base[p:p + L])
        else:
            out.append(_rand_seq(rng, L, alphabet))
    return out


@pytest.mark.parametrize("k", [2, 3, 5, 9, 13, 31, 33, 63, 70, 127])
def test_overlap_path_random_vs_oracle(ctx, oracle, k):
    rng = random.Random(500 + k)
    for n in (1, 2, 7, 60, 700, 5000):
        strs = _random_simplitig_set(rng, k, n, "ACGT" if rng.random() < 0.7 else "AC")
        s = kc.simplitigs_from_strings(ctx, strs, k)
        assert s.strings() == strs
        for complements in (True, False):
            for lb in (False, True):
                want = oracle.overlap_path(strs, k, complements, lb)
                got = kc.OverlapHamiltonianPath(s, complements, lb)
                assert got == want, (k, n, complements, lb)
        s.close()


# ------------------------------------------------------------------------------------------------ C2/C3 (T-A)
def test_superstring_from_path_reference_kats(ctx):
    # tests/global_unittest.h:11-51
    for ef, ov, strs, k, want, complements in [
        ([2, 0, NONE], [1, 2, 255], ["ACG", "TAC", "GGC"], 3, "TAcGgc", False),
        ([4, 3, 1, NONE, 5, NONE], [1, 2, 1, 255, 2, 255], ["GCC", "ACG", "TAC"], 3, "GcCGta", True),
        ([3, 1, NONE, NONE], [1, 1, 255, 255], ["GCC", "TACG"], 3, "GcCGta", True),
    ]:
        s = kc.simplitigs_from_strings(ctx, strs, k)
        out, _, L = kc.SuperstringFromPath(s, ef, ov, complements)
        assert out == want and L == len(want)
        s.close()


def test_global_reference_kats(ctx):
    # tests/global_unittest.h:113-138
    for want, k, strs, complements in [
        ("TACgt", 3, ["CGT", "TAC", "ACG"], False), ("ACGT", 1, ["ACGT"], False), ("ACgTtt", 3, ["CGT", "TTT", "ACG"], False),
        ("ACgTtt", 3, ["ACGT", "TTT"], False), ("TActt", 4, ["TACT", "ACTT"], False),
        ("TActTaaGgac", 4, ["TACT", "ACTT", "GGAC", "TAAG"], False),
        ("TTtcttttttttttttttttttttttttttga", 31, ["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTG", "TTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"], False),
        ("AtTTgtt", 4, ["ACAA", "ATTT", "AACA"], True),
    ]:
        s = kc.simplitigs_from_strings(ctx, strs, k)
        out, _, L = kc.Global(s, complements)
        assert out == want and L == len(want)
        s.close()


@pytest.mark.parametrize("k", [3, 5, 13, 31, 63, 127])
def test_global_random_with_mask_vs_oracle(ctx, oracle, k):
    rng = random.Random(900 + k)
    for n in (1, 3, 40, 1500):
        strs = _random_simplitig_set(rng, k, n, "ACGT")
        for complements in (True, False):
            rc, want, want_mask = oracle.global_(strs, k, complements, want_mask=True)
            assert rc == 0
            s = kc.simplitigs_from_strings(ctx, strs, k)
            out, mask, L = kc.Global(s, complements, want_mask=True)
            assert out == want and mask == want_mask and L == len(want)
            s.close()


def test_sparse_branch_vs_oracle(ctx, oracle):
    # D1: GlobalSparse over PartialPreSort'ed k-mers == Global over the 1-k-mer simplitigs kc_simplitigs_to_sparse builds
    rng = random.Random(4)
    for k in (3, 8, 31, 40):
        strs = _random_simplitig_set(rng, k, 300, "ACGT")
        kmers = [oracle_lib.encode_kmer(s[i:i + k]) for s in strs for i in range(len(s) - k + 1)]
        for complements in (True, False):
            rc, want, want_mask = oracle.global_sparse(kmers, k, complements, presort=True, want_mask=True)
            s = kc.simplitigs_from_strings(ctx, strs, k)
            sp = s.to_sparse()
            out, mask, _ = kc.Global(sp, complements, want_mask=True)
            assert out == want and mask == want_mask
            sp.close(); s.close()


def test_lower_bound_vs_oracle(ctx, oracle):
    rng = random.Random(11)
    for k in (3, 7, 31):
        strs = _random_simplitig_set(rng, k, 200, "ACGT")
        s = kc.simplitigs_from_strings(ctx, strs, k)
        for complements in (True, False):
            assert kc.LowerBoundLength(s, complements) == oracle.lower_bound(strs, k, complements)
        s.close()


# ------------------------------------------------------------------------------------------------ compute, goldens
def _golden(golden_dir):
    with open(os.path.join(golden_dir, "golden.json")) as f:
        return json.load(f)


def test_compute_S_golden_byte_exact(ctx, golden_dir):
    # `compute -k 31 -S data/simplitigs-k31.fa` is hash-layout independent => must equal the reference byte for byte
    g = _golden(golden_dir)["simplitigs-k31"]
    r = kc.compute_file(ctx, os.path.join(golden_dir, "simplitigs-k31.fa.gz"), 31, True, want_mask=True, assume_simplitigs=True)
    out, mask = r.strings(want_mask=True)
    assert r.stats["n_simplitigs"] == g["S"] and r.length == g["L"] and r.stats["n_kmers"] == g["N"]
    assert hashlib.md5(out.encode()).hexdigest() == g["md5"]
    assert hashlib.md5(mask.encode()).hexdigest() == g["md5_mask"]
    r.close()


@pytest.mark.parametrize("k", [13, 31, 63, 127])
@pytest.mark.parametrize("complements", [True, False])
def test_compute_spneumoniae(ctx, oracle, golden_dir, k, complements):
    path = os.path.join(golden_dir, "spneumoniae.fa.gz")
    g = _golden(golden_dir)["spneumoniae"]["%d_%d" % (k, int(complements))]
    r = kc.compute_file(ctx, path, k, complements, want_mask=True)
    out, mask = r.strings(want_mask=True)
    # T-B: identical N, identical k-mer set, min-one mask, -M == maskopt max-one of -o
    assert r.stats["n_kmers"] == g["N"]
    got_set, ones = oracle_lib.kmers_of_masked_superstring(out, k, complements)
    assert ones == g["N"] and len(got_set) == g["N"]
    mset, mones = oracle_lib.kmers_of_masked_superstring(mask, k, complements)
    assert mset == got_set
    rc, want_mask = oracle.maskopt(out, k, complements, minimize=False)
    assert rc == 0 and mask == want_mask
    assert hashlib.md5("".join(sorted(got_set)).encode()).hexdigest() == g["set_md5"]
    # Superstring length and simplitig count vs the reference.  The simplitig sets differ: the canonical rule is a
    # matching of ports, so it can close a cycle, which the reference's delete-as-you-go walk never does -- it would have run on
    # into a neighbouring chain.  A cycle is therefore opened at the port with the most preferred edge to a chain end and attached
    # to that chain (kc_cycles_decide); only a cycle without such an edge becomes a simplitig of its own.  Measured on the eight
    # runs: dL = -11, -5, -124, -39, -128, -162, -161, -392 characters, dS = -1, -1, 0, +1, -4, -2, 0, -2.
    cycles = r.stats["cycles_cut"]
    assert r.stats["n_simplitigs"] <= g["S"] + (1 if (k, complements) == (31, True) else 0), (r.stats["n_simplitigs"], g["S"], cycles)
    assert r.length <= g["L"], (r.length, g["L"])
    assert (g["L"] - r.length) / g["L"] <= 2.7e-4
    # T-C: byte-identical to the oracle's canonical-order mode
    rc, want, wmask, st = oracle.compute_file(path, k, complements, False, mode=1)
    assert rc == 0 and out == want and mask == wmask
    assert r.stats["n_simplitigs"] == st[1]
    r.close()


def test_compute_stress_reads_all_modes(ctx, oracle, tmp_path):
    reads = gen.stress_reads(20000, 400, 150, seed=3)
    fa = tmp_path / "reads.fa"
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    for k in (11, 31, 45):
        for complements in (True, False):
            r = kc.compute_file(ctx, str(fa), k, complements, want_mask=True)
            out, mask = r.strings(want_mask=True)
            rc, want, wmask, st = oracle.compute_file(str(fa), k, complements, False, mode=1)
            assert (out, mask) == (want, wmask)
            assert r.stats["n_kmers"] == st[0]
            r.close()


def test_compute_sparse_branch_end_to_end(ctx, oracle, tmp_path):
    # many isolated k-mers: S*5 >= N triggers the GlobalSparse branch (main.cpp:175-181)
    rng = random.Random(9)
    fa = tmp_path / "frag.fa"
    fa.write_text("".join(">%d\n%s\n" % (i, _rand_seq(rng, 31 + rng.choice([0, 0, 1]))) for i in range(500)))
    r = kc.compute_file(ctx, str(fa), 31, True, want_mask=True)
    assert r.stats["sparse_branch"] == 1
    out, mask = r.strings(want_mask=True)
    rc, want, wmask, st = oracle.compute_file(str(fa), 31, True, False, mode=1)
    assert st[3] == 1 and (out, mask) == (want, wmask)
    r.close()


# ------------------------------------------------------------------------------------------------ next row 1: maskopt
def test_maskopt_ones_vs_oracle(ctx, oracle, golden_dir):
    rng = random.Random(21)
    for k in (2, 5, 31, 63):
        for _ in range(5):
            up = _rand_seq(rng, 400)
            ms = "".join(c if rng.random() < 0.4 else c.lower() for c in up)
            for complements in (True, False):
                for alg in ("max-one", "min-one"):
                    rc, want = oracle.maskopt(ms, k, complements, alg == "min-one")
                    assert rc == 0
                    assert kc.Optimize(ctx, ms, k, complements, alg) == want
    with pytest.raises(kc.KcError) as e:
        kc.Optimize(ctx, "ACGTNACGTACGT", 3, True, "max-one")
    assert e.value.code == kc.api.KC_E_INVALID_CHAR


@pytest.mark.parametrize("k", [13, 31, 63, 127])
def test_maskopt_on_reference_fixtures(ctx, oracle, golden_dir, k):
    # data/global-k{K}c.fa: older-version masked superstrings kept as maskopt inputs (Makefile:18-22)
    recs = kc.read_records(os.path.join(golden_dir, "global-k%dc.fa.gz" % k))
    assert len(recs) == 1
    ms = recs[0][2].decode()
    g = _golden(golden_dir)["maskopt"][str(k)]
    for alg in ("max-one", "min-one"):
        got = kc.Optimize(ctx, ms, k, True, alg)
        assert hashlib.md5(got.encode()).hexdigest() == g[alg]


# ------------------------------------------------------------------------------------------------ next row 2: -z
def test_min_frequency_filter(ctx, oracle, golden_dir, tmp_path):
    tfa = os.path.join(golden_dir, "test.fa")
    for k, c, z, want in [(3, True, 2, 6), (3, False, 2, 5), (3, True, 3, 2), (4, True, 2, 3), (1, False, 5, 4), (1, False, 6, 2), (1, False, 10, 1)]:
        s = kc.ReadKMersFiltered(ctx, tfa, k, c, z)   # tests/parser_unittest.h:64-91
        assert s.size() == want
        assert sorted(s.keys()) == oracle.file_kmer_set_z(tfa, k, c, z)
        s.close()
    reads = gen.stress_reads(3000, 300, 100, seed=5)
    fa = tmp_path / "cov.fa"
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    # both routes of kc_compute's -z: the LDS counting build of the sharded flow on one rank (default for k >= 18) and the direct counting
    # insert (KC_Z_DIRECT=1; k < 18 takes it anyway) -- each against the oracle
    for direct in ("", "1"):
        if direct:
            os.environ["KC_Z_DIRECT"] = "1"
        else:
            os.environ.pop("KC_Z_DIRECT", None)
        try:
            for k in (9, 31, 40):
                for c in (True, False):
                    for z in (2, 3, 7):
                        rc, want, wmask, st = oracle.compute_file_z(str(fa), k, c, 1, z)
                        if rc != 0:
                            with pytest.raises(kc.KcError):
                                kc.compute_file(ctx, str(fa), k, c, want_mask=True, min_frequency=z)
                            continue
                        r = kc.compute_file(ctx, str(fa), k, c, want_mask=True, min_frequency=z)
                        out, mask = r.strings(want_mask=True)
                        assert (out, mask) == (want, wmask) and r.stats["n_kmers"] == st[0], (direct, k, c, z)
                        r.close()
        finally:
            os.environ.pop("KC_Z_DIRECT", None)
    # growing while counting (many small adds) keeps the counters
    s = kc.KmerSet(ctx, 21, True)
    ctx.check(ctx.lib.kc_set_enable_counting(s.h))
    for r in reads:
        s.AddKMers(r)
    ctx.check(ctx.lib.kc_set_filter_min_frequency(s.h, 3))
    assert sorted(s.keys()) == oracle.file_kmer_set_z(str(fa), 21, True, 3)
    s.close()
