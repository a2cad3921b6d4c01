"""Pins the oracle restatement against the reference's own headers compiled in place (oracle/_ref).  Runs where
oracle/_ref exists (this container; on the GPU box the prebuilt oracle/_ref travels with the repo snapshot).  Not gpu."""
import os
import random

import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _rand(rng, n, a="ACGT"):
    return "".join(rng.choice(a) for _ in range(n))


def test_primitives_match_reference(oracle, reference):
    rng = random.Random(1)
    for k in (1, 2, 7, 31, 32, 45, 63, 64, 99, 127):
        for _ in range(30):
            x = rng.getrandbits(2 * k)
            assert oracle.reverse_complement(x, k) == reference.reverse_complement(x, k)
            d = rng.randint(0, k)
            assert oracle.bit_prefix(x, k, d) == reference.bit_prefix(x, k, d)
            assert oracle.bit_suffix(x, k, d) == reference.bit_suffix(x, k, d)
            assert oracle.hash(x, k) == reference.hash(x, k)
    for c in "ACGTacgtNnXx\n*":
        assert oracle.nucleotide_to_int(c) == reference.nucleotide_to_int(c)
        for m in (0, 1):
            if c.isalpha():
                assert oracle.masked(c, m) == reference.masked(c, m)


def test_kmer_sets_match_reference(oracle, reference):
    rng = random.Random(2)
    for k in (1, 3, 11, 31, 32, 63, 64, 127):
        for _ in range(6):
            seq = _rand(rng, rng.choice([0, k - 1, k, 5 * k, 700]), "ACGTacgtN" if rng.random() < 0.5 else "ACGT")
            for c in (True, False):
                for cs in (True, False):
                    assert oracle.kmer_set(seq, k, c, cs) == reference.kmer_set(seq, k, c, cs)


def test_reference_order_simplitigs_match_reference(oracle, reference):
    # the khash layout emulation: identical simplitigs in identical order and orientation
    rng = random.Random(3)
    for k in (3, 5, 12, 31, 33, 63, 80, 127):
        for _ in range(5):
            seq = _rand(rng, rng.choice([k, 300, 3000]), "ACGT" if rng.random() < 0.6 else "AC")
            for c in (True, False):
                assert oracle.simplitigs_of_sequence(seq, k, c, mode=0) == reference.simplitigs_of_sequence(seq, k, c)


def test_overlap_superstring_global_match_reference(oracle, reference):
    rng = random.Random(4)
    for k in (2, 3, 5, 9, 31, 33, 63, 70, 127):
        base = _rand(rng, max(k, 500))
        for n in (1, 2, 9, 150):
            strs = []
            for _ in range(n):
                L = k + rng.choice([0, 0, 1, 3, 30])
                p = rng.randrange(0, len(base) - L + 1)
                strs.append(base[p:p + L] if rng.random() < 0.6 else _rand(rng, L))
            for c in (True, False):
                for lb in (False, True):
                    assert oracle.overlap_path(strs, k, c, lb) == reference.overlap_path(strs, k, c, lb)
                ef, ov = reference.overlap_path(strs, k, c, False)
                assert oracle.superstring_from_path(strs, ef, ov, k, c, True) == reference.superstring_from_path(strs, ef, ov, k, c, True)
                assert oracle.global_(strs, k, c, True) == reference.global_(strs, k, c, True)
                assert oracle.lower_bound(strs, k, c) == reference.lower_bound(strs, k, c)
            kmers = sorted(set(oracle_lib.encode_kmer(s[i:i + k]) for s in strs for i in range(len(s) - k + 1)))
            rng.shuffle(kmers)
            for c in (True, False):
                assert oracle.global_sparse(kmers, k, c, True, True) == reference.global_sparse(kmers, k, c, True, True)
    assert reference.global_([], 3, True)[0] == -2 and oracle.global_([], 3, True)[0] == -2


@pytest.mark.parametrize("k,c", [(13, 0), (31, 0), (63, 1), (127, 0)])
def test_compute_file_byte_identical_to_reference(oracle, reference, k, c, tmp_path):
    import gzip
    p = tmp_path / "sp.fa"
    p.write_bytes(gzip.open(os.path.join(GOLDEN, "spneumoniae.fa.gz")).read())
    rc, want, wmask, wst = reference.compute_file(str(p), k, bool(c), False, True, tmpdir=str(tmp_path))
    rc2, out, mask, st = oracle.compute_file(str(p), k, bool(c), False, mode=0)
    assert rc == 0 and rc2 == 0 and out == want and mask == wmask and st[:4] == wst[:4]


def test_sparse_branch_end_to_end_matches_reference(oracle, reference, tmp_path):
    rng = random.Random(6)
    fa = tmp_path / "frag.fa"
    fa.write_text("".join(">%d\n%s\n" % (i, _rand(rng, 31 + rng.choice([0, 0, 1]))) for i in range(400)))
    rc, want, wmask, wst = reference.compute_file(str(fa), 31, True, False, True, tmpdir=str(tmp_path))
    rc2, out, mask, st = oracle.compute_file(str(fa), 31, True, False, mode=0)
    assert wst[3] == 1 and st[3] == 1 and (out, mask) == (want, wmask)


def test_min_frequency_filter_matches_reference(oracle, reference, tmp_path):
    # next row 2: -z (ReadKMersFiltered, parser.h:122-141); KATs of tests/parser_unittest.h:64-91 + end-to-end bytes
    tfa = os.path.join(GOLDEN, "test.fa")
    for k, c, z, want in [(3, True, 2, 6), (3, False, 2, 5), (3, True, 3, 2), (4, True, 2, 3), (1, False, 5, 4), (1, False, 6, 2), (1, False, 10, 1)]:
        assert reference.read_kmers_filtered_count(tfa, k, c, z) == want
        assert len(oracle.file_kmer_set_z(tfa, k, c, z)) == want
    import gen
    reads = gen.stress_reads(3000, 300, 100, seed=5)
    fa = tmp_path / "cov.fa"
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    for k in (9, 31, 40):
        for c in (True, False):
            for z in (2, 3, 7):
                rc, want, wmask, wst = reference.compute_file_z(str(fa), k, c, z, True, tmpdir=str(tmp_path))
                rc2, out, mask, st = oracle.compute_file_z(str(fa), k, c, 0, z)
                assert rc == rc2
                if rc == 0:
                    assert (out, mask, st[:3]) == (want, wmask, wst[:3])
