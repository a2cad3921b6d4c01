"""Format converters of the CLI (ms2mssep / mssep2ms / ms2spss / spss2ms, reference conversions.h:16-91, main.cpp:445-667)
against the reference's own conversions.h compiled in place (oracle/_ref) and the committed golden digests.  Host-side
code only: no GPU is touched, so everything here runs under -m "not gpu"."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

import oracle_lib
from test_cli import BIN, GOLDEN, run

GOLD = json.load(open(os.path.join(GOLDEN, "golden.json")))


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def unz(tmp_path, name):
    p = os.path.join(str(tmp_path), name[:-3])
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return p


def write(tmp_path, name, text):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(text)
    return p


EDGE_MS = {
    "empty.fa": "",
    "noseq.fa": ">only a header\n",
    "lower.fa": ">x\nacgtacgt\n",
    "upper_to_end.fa": ">x\nacGTAC\n",            # a run of ones reaching the end: no tail, no line end
    "upper_in_tail.fa": ">x\nACgTacGgtacca\n",      # ones inside the k-1 tail of the previous run
    "multiline.fa": ">x c\nACGTac\ngtACgt\nac\n",
    "two_records.fa": ">a\nACgt\n>b\nTTtt\n",       # only the first record is read
    "nonacgt.fa": ">x\nACNnXx01acgt\n",
    "fastq.fq": "@r\nACGtacg\n+\nIIIIIII\n",
}
EDGE_SPSS = {
    "empty.fa": "",
    "short.fa": ">0\nACG\n>1\nAC\n",
    "mixed.fa": ">0\nACGTA\n>1\nAC\n>2\nacgtacgt\n>3\nACGTACG\n",
    "multiline.fa": ">0\nACG\nTAC\n>1\nGGGGG\n",
}


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built")
@pytest.mark.parametrize("k", [1, 2, 4, 31])
def test_converters_match_reference_on_edge_cases(tmp_path, k):
    ref = oracle_lib.Reference()
    for name, text in EDGE_MS.items():
        p = write(tmp_path, name, text)
        m, s = os.path.join(str(tmp_path), "m.txt"), os.path.join(str(tmp_path), "s.txt")
        r = run("ms2mssep", "-m", m, "-s", s, p)
        assert r.returncode == 0, r.stderr
        assert (open(s).read(), open(m).read()) == ref.split_ms(p), name
        r = run("ms2spss", "-k", str(k), p)
        assert r.returncode == 0 and r.stdout == ref.ms_to_spss(p, k), (name, k)
    for name, text in EDGE_SPSS.items():
        p = write(tmp_path, "spss_" + name, text)
        r = run("spss2ms", "-k", str(k), p)
        assert r.returncode == 0 and r.stdout == ref.spss_to_ms(p, k), (name, k)


@pytest.mark.skipif(not oracle_lib.have_reference(), reason="oracle/_ref not built")
def test_join_matches_reference(tmp_path):
    ref = oracle_lib.Reference()
    for sup, mask in [("ACGTACGT", "10100000"), ("acgtACGT", "11110000"), ("ACGT", "1111"), ("", ""), ("ACGT  \n", "0110\n"),
                      ("ACGT extra", "1001 tail"), ("ACNX", "0101")]:
        s, m, o = write(tmp_path, "s.txt", sup), write(tmp_path, "m.txt", mask), os.path.join(str(tmp_path), "o.fa")
        r = run("mssep2ms", "-s", s, "-m", m, "-o", o)
        assert r.returncode == 0, r.stderr
        assert open(o).read() == ref.join_ms(sup, mask), (sup, mask)
    # superstring from stdin
    r = subprocess.run([BIN, "mssep2ms", "-m", write(tmp_path, "m2.txt", "0110")], input="ACGT\n", capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == ">superstring\naCGt\n"
    # a mask shorter than the superstring counts as zeros from its end on
    r = subprocess.run([BIN, "mssep2ms", "-m", write(tmp_path, "m3.txt", "11")], input="ACGT\n", capture_output=True, text=True)
    assert r.stdout == ">superstring\nACgt\n"


@pytest.mark.parametrize("k", [13, 31, 63, 127])
def test_converters_on_reference_data(tmp_path, k):
    """The reference's own masked superstrings (tests/testdata global-k*c.fa): digests pinned by make_golden.py against
    conversions.h compiled in place, plus the round trips the formats promise."""
    p = unz(tmp_path, "global-k%dc.fa.gz" % k)
    g = GOLD["converters"][str(k)]
    m, s = os.path.join(str(tmp_path), "m.txt"), os.path.join(str(tmp_path), "s.txt")
    assert run("ms2mssep", "-m", m, "-s", s, p).returncode == 0
    sup, mask = open(s).read(), open(m).read()
    assert md5(sup) == g["superstring"] and md5(mask) == g["mask"]
    o = os.path.join(str(tmp_path), "joined.fa")
    assert run("mssep2ms", "-m", m, "-s", s, "-o", o).returncode == 0
    joined = open(o).read()
    assert md5(joined) == g["joined"]
    orc = oracle_lib.Oracle()
    ms = orc.parse_records(p)[0][2]
    assert joined == ">superstring\n" + ms + "\n"                       # split then join gives the sequence back
    spss = os.path.join(str(tmp_path), "spss.fa")
    assert run("ms2spss", "-k", str(k), "-o", spss, p).returncode == 0
    assert md5(open(spss).read()) == g["spss"]
    back = run("spss2ms", "-k", str(k), spss)
    assert back.returncode == 0 and back.stdout.startswith(">superstring " + spss + "\n")
    # ms -> rSPSS -> ms represents the same k-mer set
    ms2 = back.stdout.split("\n")[1]
    a, _ = oracle_lib.kmers_of_masked_superstring(ms, k, True)
    b, _ = oracle_lib.kmers_of_masked_superstring(ms2, k, True)
    assert a == b and len(a) == GOLD["maskopt"][str(k)]["N"]
    assert md5(ms2) == g["spss_ms_seq"]


def test_spss_fixture_and_simplitigs(tmp_path):
    """spss.fa (reference tests/testdata) and the simplitig file: spss2ms output equals reading the records directly."""
    for name, k in (("spss.fa", 3), ("spss.fa", 5)):
        r = run("spss2ms", "-k", str(k), os.path.join(GOLDEN, name))
        assert r.returncode == 0
        assert md5(r.stdout.split("\n", 1)[1]) == GOLD["converters"]["spss.fa"][str(k)]
    p = unz(tmp_path, "simplitigs-k31.fa.gz")
    r = run("spss2ms", "-k", "31", p)
    recs = oracle_lib.Oracle().parse_records(p)
    seq = r.stdout.split("\n")[1]
    assert len(seq) == sum(len(x[2]) for x in recs)
    kset, ones = oracle_lib.kmers_of_masked_superstring(seq, 31, True)
    assert ones == sum(len(x[2]) - 30 for x in recs) and len(kset) == GOLD["simplitigs-k31"]["N"]


def test_reference_unit_test_vectors(tmp_path):
    """The known answers of the reference's tests/conversions_unittest.h:12-96 (runstest.fa, spss.fa, JoinMS)."""
    runs, spss = os.path.join(GOLDEN, "runstest.fa"), os.path.join(GOLDEN, "spss.fa")
    m, s = os.path.join(str(tmp_path), "m.txt"), os.path.join(str(tmp_path), "s.txt")
    assert run("ms2mssep", "-m", m, "-s", s, runs).returncode == 0
    assert open(m).read() == "10011100010110\n" and open(s).read() == "ACGCGTTACGTATT\n"
    for k, want in [(2, ">0\nAC\n>1\nCGTT\n>2\nGT\n>3\nATT\n"), (3, ">0\nACG\n>1\nCGTTA\n>2\nGTA\n>3\nATT\n")]:
        assert run("ms2spss", "-k", str(k), runs).stdout == want
    for k, want in [(1, "ACCCGAACCGTAATGCACCCGTTTAACGA"), (2, "ACCCGAAcCGTaATGcACCCGTTTAACg"), (3, "ACCCGAacCGtaATgcACCCGTTTAAcg"),
                    (4, "ACCCGaacCgtaAtgcACCCGTTTAacg"), (5, "ACCCgaacACCCGTTTaacg")]:
        assert run("spss2ms", "-k", str(k), spss).stdout == ">superstring " + spss + "\n" + want + "\n"
    for mask, sup, want in [("1001101\n", "ACGTTTA\n", ">superstring\nAcgTTtA\n"), ("0111110000\n", "CCCGGTATTA\n", ">superstring\ncCCGGTatta\n")]:
        r = run("mssep2ms", "-m", write(tmp_path, "jm.txt", mask), "-s", write(tmp_path, "js.txt", sup))
        assert r.stdout == want


def test_converter_cli_validation():
    # main.cpp:445-667 messages; `-s` alone is not a redirection for ms2mssep in the reference either
    for args, msg in [
        (["ms2mssep", "-m", "/tmp/kc_m.txt"], "Usage:"),
        (["ms2mssep", "x.fa"], "Cannot have both superstring and mask redirected to stdout."),
        (["ms2mssep", "-s", "/tmp/kc_s.txt", "x.fa"], "Cannot have both superstring and mask redirected to stdout."),
        (["ms2mssep", "-m", "/tmp/kc_m.txt", "-m", "/tmp/kc_m2.txt", "x.fa"], "Error: -m parameter provided multiple times"),
        (["mssep2ms"], "Cannot have both superstring and mask redirected from stdin."),
        (["mssep2ms", "-s", "/tmp/kc_s.txt", "-s", "/tmp/kc_s.txt"], "Error: -s parameter provided multiple times"),
        (["ms2spss", "x.fa"], "Required parameter k not set."),
        (["ms2spss", "-k", "-1", "x.fa"], "k must be positive."),
        (["ms2spss"], "Required positional parameter path to the file not set."),
        (["spss2ms", "-k", "abc", "x.fa"], "Usage:"),
        (["spss2ms", "x.fa"], "Required parameter k not set."),
    ]:
        p = run(*args)
        assert p.returncode == 1 and msg in p.stderr, (args, p.stderr)
    for sub, pos in (("ms2mssep", "<ms>"), ("ms2spss", "<ms>"), ("spss2ms", "<fasta>"), ("mssep2ms", "[options]\n")):
        p = run(sub, "-h")
        assert p.returncode == 0 and "Usage:   kmercamel " + sub + " [options]" in p.stderr and pos in p.stderr
    p = run("spss2ms", "-k", "3", "/nonexistent/file.fa")
    assert p.returncode != 0
