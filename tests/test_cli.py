"""CLI tests: argument validation runs anywhere (no GPU is touched before validation ends); the functional part is -m gpu."""
import hashlib
import json
import os
import subprocess

import pytest
import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
SHIM = os.path.join(ROOT, "kmercamel_amd", "bin", "shim_check")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "kmercamel_amd")])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "kmercamel_amd", "src")])


def run(*args):
    _build()
    return subprocess.run([BIN] + list(args), capture_output=True, text=True)


def test_cli_usage_and_validation_messages():
    # main.cpp:277-308, 669-691: message on stderr + usage, exit code 1; -h exits 0
    assert run().returncode == 0
    assert run("-h").returncode == 0
    p = run("frobnicate")
    assert p.returncode == 1 and "Command 'frobnicate' not recognized." in p.stderr
    p = run("compute", "-h")
    assert p.returncode == 0 and "Usage:   kmercamel compute [options] <fasta>" in p.stderr
    for args, msg in [
        (["compute"], "Required positional parameter path to the file not set."),
        (["compute", "x.fa"], "Required parameter k not set."),
        (["compute", "-k", "-3", "x.fa"], "k must be positive."),
        (["compute", "-k", "128", "x.fa"], "k > 127 not supported for the algorithm 'greedy'."),
        (["compute", "-k", "31", "-d", "3", "x.fa"], "Unsupported argument d for algorithm 'greedy'."),
        (["compute", "-k", "31", "-a", "streaming", "-M", "m.fa", "x.fa"], "Outputting mask maximizing number of ones is possible only with greedy."),
        (["compute", "-k", "31", "-a", "local", "-S", "x.fa"], "Optimization for the input being simplitigs is possible only with greedy."),
        (["compute", "-k", "31", "-z", "0", "x.fa"], "Minimum frequency '-z' must be between 1 and 255."),
        (["lowerbound", "-k", "31", "-z", "256", "x.fa"], "Minimum frequency '-z' must be between 1 and 255."),
        (["compute", "-k", "31", "-z", "300", "x.fa"], "Minimum frequency '-z' must be between 1 and 255."),
        (["compute", "-k", "31", "-z", "2", "-S", "x.fa"], "Inputting simplitigs is not compatible with frequency filterring."),
        (["compute", "-k", "abc", "x.fa"], "Usage:"),
        (["maskopt", "x.fa"], "Required parameter k not set."),
        (["maskopt", "-k", "31", "-t", "bogus", "x.fa"], "Algorithm 'bogus' not recognized."),
        (["lowerbound", "-k", "0", "x.fa"], "Required parameter k not set."),
    ]:
        p = run(*args)
        assert p.returncode == 1 and msg in p.stderr, (args, p.stderr)
    if os.path.exists("m.fa"):
        os.remove("m.fa")


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


@pytest.mark.gpu
def test_cli_compute_S_golden(tmp_path):
    g = json.load(open(os.path.join(GOLDEN, "golden.json")))["simplitigs-k31"]
    out, mask = str(tmp_path / "o.fa"), str(tmp_path / "m.fa")
    src = os.path.join(GOLDEN, "simplitigs-k31.fa.gz")
    p = run("compute", "-k", "31", "-S", "-o", out, "-M", mask, src)
    assert p.returncode == 0, p.stderr
    lines = open(out).read().split("\n")
    assert lines[0] == ">maskedsuperstring dataset='%s' k=31 alg=greedy mask=min-one mode=bidirectional" % src
    assert md5(lines[1]) == g["md5"] and lines[2] == "" and len(lines) == 3
    mlines = open(mask).read().split("\n")
    assert mlines[0].endswith("mask=max-one mode=bidirectional") and md5(mlines[1]) == g["md5_mask"]
    for msg in ("Started computation of a masked superstring from", "Finished 1. part: simplitigs (%d simplitigs)." % g["S"],
                "Finished 2. part: Hamiltonian path.", "Finished 3. part: masked superstring (l=%d)." % g["L"]):
        assert msg in p.stderr


@pytest.mark.gpu
def test_cli_compute_raw_fasta_and_maskopt_and_lowerbound(tmp_path, oracle):
    src = os.path.join(GOLDEN, "spneumoniae.fa.gz")
    g = json.load(open(os.path.join(GOLDEN, "golden.json")))
    for k, flag in ((31, []), (63, ["-u"])):
        c = not flag
        p = run("compute", "-k", str(k), *flag, src)
        assert p.returncode == 0, p.stderr
        lines = p.stdout.split("\n")
        assert lines[0] == ">maskedsuperstring dataset='%s' k=%d alg=greedy mask=min-one mode=%s" % (src, k, "bidirectional" if c else "unidirectional")
        rc, want, _, st = oracle.compute_file(src, k, c, False, mode=1, want_mask=False)
        assert lines[1] == want
        assert "Finished collecting k-mers: %d %d-mers." % (g["spneumoniae"]["%d_%d" % (k, int(c))]["N"], k) in p.stderr
    # maskopt on the reference's fixture (Makefile:18-22)
    ms = os.path.join(GOLDEN, "global-k31c.fa.gz")
    for alg in ("max-one", "min-one"):
        p = run("maskopt", "-k", "31", "-t", alg, ms)
        assert p.returncode == 0, p.stderr
        lines = p.stdout.split("\n")
        assert " reoptimized=%s" % alg in lines[0] and md5(lines[1]) == g["maskopt"]["31"][alg]
    # lowerbound -S
    recs = oracle.parse_records(os.path.join(GOLDEN, "simplitigs-k31.fa.gz"))
    want = oracle.lower_bound([r[2] for r in recs], 31, True)
    p = run("lowerbound", "-k", "31", "-S", os.path.join(GOLDEN, "simplitigs-k31.fa.gz"))
    assert p.returncode == 0 and p.stdout.strip() == str(want)
    # -z 2 on a file where every record is duplicated once
    dup = tmp_path / "dup.fa"
    dup.write_text(">a\nACGTTGCAAGGCTTACCGATAGGCTAGCTAGGATCGA\n>b\nACGTTGCAAGGCTTACCGATAGGCTAGCTAGGATCGA\n>c\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTGGGG\n")
    p = run("compute", "-k", "11", "-z", "2", str(dup))
    rc, want, _, st = oracle.compute_file_z(str(dup), 11, True, 1, 2, want_mask=False)
    assert p.returncode == 0 and p.stdout.split("\n")[1] == want
    # no k-mers (main.cpp:155-158)
    fa = tmp_path / "n.fa"
    fa.write_text(">x\nACGT\n")
    p = run("compute", "-k", "31", str(fa))
    assert p.returncode == 1 and "contains no k-mers" in p.stderr


@pytest.mark.gpu
def test_dropin_templates_all_key_widths():
    _build()
    p = subprocess.run([SHIM, os.path.join(GOLDEN, "test.fa")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("PASS") == 3


def test_fast_fasta_loader_matches_tokenizer(tmp_path):
    """tests/host/parse_check.cpp: the CLI's one-pass FASTA loader (join_fasta) vs the kseq-semantics tokenizer it
    shortcuts, on edge cases and 200 000 random byte soups (host-only)."""
    exe = str(tmp_path / "parse_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "host", "parse_check.cpp"),
                           "-I" + os.path.join(ROOT, "kmercamel_amd", "src"), "-lz", "-pthread", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 0 and "parse_check ok" in p.stdout, p.stdout


@pytest.mark.gpu
def test_cli_streamed_fasta_tokeniser_equals_host_parser(tmp_path):
    """compute reads FASTA through the device-side tokeniser (kc_read_fasta_device: pinned chunks, header lines and line breaks
    cut out on the GPU); KC_CLI_NOSTREAM=1 takes the host tokenizer with kseq's semantics (kseq.h:183-224).  Same bytes out on
    awkward inputs: multi-line records, CRLF, empty lines, junk before the first header, '>' inside a line, header at the very
    end, no trailing newline, '@' headers, gzip, stdin, and random mixtures of all of that."""
    import gzip
    import random
    rng = random.Random(17)

    def seq(n):
        return "".join(rng.choice("ACGTacgtNn") if rng.random() < 0.03 else rng.choice("ACGT") for _ in range(n))

    cases = {
        "multiline.fa": ">a desc\n" + "\n".join(seq(60) for _ in range(40)) + "\n>b\n" + seq(75) + "\n" + seq(3) + "\n",
        "crlf.fa": ">a\r\n" + "\r\n".join(seq(50) for _ in range(30)) + "\r\n>b\r\n" + seq(200) + "\r\n",
        "blank_lines.fa": "\n\n>a\n\n" + seq(100) + "\n\n\n" + seq(100) + "\n>b\n\n>c\n" + seq(150),
        "junk_before.fa": "this is not fasta\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>a\n" + seq(300) + "\n",
        "gt_inside.fa": ">a\n" + seq(80) + ">" + seq(80) + "\n" + seq(40) + "@" + seq(40) + "\n>b\n" + seq(120) + "\n",
        "header_last.fa": ">a\n" + seq(300) + "\n>",
        "at_headers.fa": ">a\n" + seq(200) + "\n@b looks like fastq but has no plus line\n" + seq(200) + "\n",
    }
    soup = []
    for r in range(300):
        soup.append((">" if rng.random() < 0.8 else "@") + "r%d %s" % (r, "x" * rng.randrange(0, 30)))
        for _ in range(rng.randrange(0, 6)):
            soup.append(seq(rng.randrange(0, 90)) + ("\r" if rng.random() < 0.1 else ""))
    cases["soup.fa"] = "\n".join(soup) + ("\n" if rng.random() < 0.5 else "")
    big = []
    for r in range(3000):   # > 32 MB: several chunks, records cut by chunk borders
        big.append(">big%d" % r)
        s = seq(12000)
        big.extend(s[i:i + 70] for i in range(0, len(s), 70))
    cases["big.fa"] = "\n".join(big) + "\n"
    for name, text in cases.items():
        path = str(tmp_path / name)
        with open(path, "w", newline="") as f:
            f.write(text)
        for k in (("31",) if name == "big.fa" else ("31", "17")):
            a = run("compute", "-k", k, path)
            env = dict(os.environ, KC_CLI_NOSTREAM="1")
            b = subprocess.run([BIN, "compute", "-k", k, path], capture_output=True, text=True, env=env)
            assert a.returncode == b.returncode == 0, (name, a.stderr[-500:], b.stderr[-500:])
            assert a.stdout == b.stdout, (name, k)
            assert a.stdout.count("\n") == 2
    # gzip and stdin go through the same reader
    path = str(tmp_path / "multiline.fa")
    with open(path, "rb") as f:
        raw = f.read()
    with gzip.open(path + ".gz", "wb") as f:
        f.write(raw)
    want = run("compute", "-k", "31", path).stdout.split("\n")[1]
    assert run("compute", "-k", "31", path + ".gz").stdout.split("\n")[1] == want
    p = subprocess.run([BIN, "compute", "-k", "31", "-"], input=raw, capture_output=True)
    assert p.returncode == 0 and p.stdout.decode().split("\n")[1] == want
    # FASTQ on stdin / through a FIFO (parser.h:88-101 + kseq.h:212-223): inputs that cannot be opened twice never go through the
    # device tokeniser (it consumes the stream before it can say "FASTQ")
    reads = [seq(150) for _ in range(200)]
    fq = "".join("@q%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)).encode()
    fa = str(tmp_path / "reads.fa")
    with open(fa, "w") as f:
        f.write("".join(">q%d\n%s\n" % (i, r) for i, r in enumerate(reads)))
    want_q = run("compute", "-k", "31", fa).stdout.split("\n")[1]
    p = subprocess.run([BIN, "compute", "-k", "31", "-"], input=fq, capture_output=True)
    assert p.returncode == 0 and p.stdout.decode().split("\n")[1] == want_q, p.stderr.decode()[-300:]
    fifo = str(tmp_path / "reads.fifo")
    os.mkfifo(fifo)
    import threading

    def feed():
        with open(fifo, "wb") as f:
            f.write(fq)
    t = threading.Thread(target=feed)
    t.start()
    p = subprocess.run([BIN, "compute", "-k", "31", fifo], capture_output=True)
    t.join()
    assert p.returncode == 0 and p.stdout.decode().split("\n")[1] == want_q, p.stderr.decode()[-300:]
    fqf = str(tmp_path / "reads.fq")
    with open(fqf, "wb") as f:
        f.write(fq)
    assert run("compute", "-k", "31", fqf).stdout.split("\n")[1] == want_q   # a regular FASTQ file: tokeniser says FASTQ, host parser re-opens it
    # -S (simplitigs_from_fasta, simplitigs.h:78-103) reads through the device tokeniser too: the records then reach kc_compute separated
    # by '\n' on the device and their extents are found there; same bytes as the host loader on multi-line / CRLF / blank-line records, with
    # and without -M, and the same complaint about a record shorter than k
    srecs = [seq(rng.randrange(31, 260)).upper().replace("N", "A") for _ in range(120)]
    stexts = {
        "s_plain.fa": "".join(">s%d\n%s\n" % (i, r) for i, r in enumerate(srecs)),
        "s_multiline_crlf.fa": "".join(">s%d some words\r\n%s\r\n" % (i, "\r\n".join(r[j:j + 50] for j in range(0, len(r), 50))) for i, r in enumerate(srecs)),
        "s_blank_and_tail.fa": "\n".join(">s%d\n\n%s\n" % (i, "\n".join(r[j:j + 70] for j in range(0, len(r), 70))) for i, r in enumerate(srecs)) + "\n>last\n" + srecs[0],
        "s_short_record.fa": ">a\n" + srecs[0] + "\n>b\nACGTACGT\n>c\n" + srecs[1] + "\n",
    }
    for name, text in stexts.items():
        sp = str(tmp_path / name)
        with open(sp, "w", newline="") as f:
            f.write(text)
        for extra in ([], ["-M", str(tmp_path / "sm.fa")]):
            a = subprocess.run([BIN, "compute", "-k", "31", "-S"] + extra + [sp], capture_output=True, text=True)
            ma = open(extra[1]).read() if extra and os.path.exists(extra[1]) else None
            b = subprocess.run([BIN, "compute", "-k", "31", "-S"] + extra + [sp], capture_output=True, text=True, env=dict(os.environ, KC_CLI_NOSTREAM="1"))
            mb = open(extra[1]).read() if extra and os.path.exists(extra[1]) else None
            assert a.returncode == b.returncode and a.stdout == b.stdout and ma == mb, (name, extra, a.stderr[-300:], b.stderr[-300:])
            assert (a.returncode == 0) == (name != "s_short_record.fa"), (name, a.returncode)
    # -o / -M files: header line, data, newline -- through the descriptor path
    o, m = str(tmp_path / "o.fa"), str(tmp_path / "m.fa")
    p = run("compute", "-k", "31", "-o", o, "-M", m, path)
    assert p.returncode == 0
    lo, lm = open(o).read().split("\n"), open(m).read().split("\n")
    assert lo[0].startswith(">maskedsuperstring") and lo[1] == want and lo[2] == "" and len(lo) == 3
    assert "mask=max-one" in lm[0] and lm[1].upper() == want.upper() and lm[2] == "" and len(lm) == 3


@pytest.mark.gpu
def test_cli_large_input_takes_the_mapped_output_path(tmp_path):
    """Inputs beyond 64 MB with `-o FILE`: the output file is fallocate'd and mapped while the GPU works, its pages are touched by
    background threads, and the result arrives through kc_result_write_mem (pinned chunks, copy threads).  Same bytes as the plain
    descriptor path (KC_CLI_NOPREALLOC=1), as stdout, and with the helpers switched off or odd (no touching threads, 3 I/O threads, one
    serial thread); the input has CRLF lines, so that a '\\r' now and then ends a 32 MB chunk and has to wait for its '\\n'."""
    import random
    rng = random.Random(5)
    path = str(tmp_path / "large.fa")
    with open(path, "w", newline="") as f:
        for r in range(6200):
            s = "".join(rng.choices("ACGT", k=12000))
            eol = "\r\n" if r % 3 == 0 else "\n"
            f.write(">rec%d%s" % (r, eol))
            f.write(eol.join(s[i:i + 61] for i in range(0, len(s), 61)) + eol)
    assert os.path.getsize(path) > (64 << 20)
    ref = run("compute", "-k", "31", path)
    assert ref.returncode == 0
    want = ref.stdout.split("\n")
    assert len(want) == 3 and want[2] == ""
    for env in ({}, {"KC_CLI_NOPREALLOC": "1"}, {"KC_CLI_TOUCH_THREADS": "0"}, {"KC_IO_THREADS": "3"}, {"KC_IO_SERIAL": "1"}, {"KC_CLI_NO_ZAP": "1"}):
        o = str(tmp_path / "o.fa")
        if os.path.exists(o):
            os.unlink(o)
        p = subprocess.run([BIN, "compute", "-k", "31", "-o", o, path], capture_output=True, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, (env, p.stderr[-400:])
        got = open(o).read().split("\n")
        assert got[0].startswith(">maskedsuperstring") and got[1] == want[1] and got[2] == "" and len(got) == 3, env
    # -M: the mask file takes the mapped path as well
    outs = []
    for env in ({}, {"KC_CLI_NOPREALLOC": "1"}):
        o2, m2 = str(tmp_path / "o2.fa"), str(tmp_path / "m2.fa")
        for f in (o2, m2):
            if os.path.exists(f):
                os.unlink(f)
        p = subprocess.run([BIN, "compute", "-k", "31", "-o", o2, "-M", m2, path], capture_output=True, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, (env, p.stderr[-400:])
        outs.append((open(o2).read(), open(m2).read()))
    assert outs[0] == outs[1] and outs[0][0].split("\n")[1] == want[1]
    ml = outs[0][1].split("\n")
    assert "mask=max-one" in ml[0] and ml[1].upper() == want[1].upper() and ml[2] == "" and len(ml) == 3
    # --gpus N on a large plain file: every rank reads its own byte range through the device tokeniser and writes its own slice of the
    # mapped output (run_compute_sharded_streamed); 2 and 3 ranks sharing the GPU, with -M, into files and to stdout
    for gp in ("2", "3"):
        o3, m3 = str(tmp_path / "o3.fa"), str(tmp_path / "m3.fa")
        for f in (o3, m3):
            if os.path.exists(f):
                os.unlink(f)
        p = subprocess.run([BIN, "compute", "-k", "31", "--gpus", gp, "-o", o3, "-M", m3, path], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
        assert p.returncode == 0, p.stderr[-600:]
        assert "every rank has its share of the input on its device" in p.stderr and "every slice is in place" in p.stderr, p.stderr[-800:]
        assert (open(o3).read(), open(m3).read()) == outs[0], gp
    p = subprocess.run([BIN, "compute", "-k", "31", "--gpus", "2", path], capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout == ref.stdout
    # a rank that fails ends the process -- and leaves EMPTY output files behind, not the preallocated 74 MB of whatever the pages held
    for f in (o3, m3):
        os.unlink(f)
    p = subprocess.run([BIN, "compute", "-k", "31", "--gpus", "2", "-o", o3, "-M", m3, path], capture_output=True, text=True,
                       env=dict(os.environ, KC_TEST_FAIL_RANK="1"))
    assert p.returncode != 0 and "kmercamel_amd: rank" in p.stderr, p.stderr[-400:]   # (rank 1's own message, or a peer's that was released: whichever leaves first)
    assert os.path.getsize(o3) == 0 and os.path.getsize(m3) == 0
    # an output file that already holds something longer is cut to the new length
    with open(o, "w") as f:
        f.write("x" * (len(want[1]) + 5000000))
    assert subprocess.run([BIN, "compute", "-k", "31", "-o", o, path], capture_output=True).returncode == 0
    assert open(o).read().split("\n")[1] == want[1]
    # `maskopt` on that 74 MB masked superstring: the streamed path (device tokeniser, kc_maskopt_device, mapped output) against the host
    # parser + kc_maskopt_ones (KC_CLI_NOSTREAM=1), both algorithms, into a file and to stdout; a comment on the header line is kept
    ms = str(tmp_path / "ms.fa")
    with open(ms, "w") as f:
        f.write(">superstring some comment here\n" + want[1] + "\n")
    for alg in ("max-one", "min-one"):
        ref = subprocess.run([BIN, "maskopt", "-t", alg, "-k", "31", ms], capture_output=True, text=True, env=dict(os.environ, KC_CLI_NOSTREAM="1"))
        assert ref.returncode == 0, ref.stderr[-300:]
        assert ref.stdout.startswith(">superstring reoptimized=%s some comment here\n" % alg)
        got = subprocess.run([BIN, "maskopt", "-t", alg, "-k", "31", ms], capture_output=True, text=True)
        assert got.returncode == 0 and got.stdout == ref.stdout, alg
        mo = str(tmp_path / "mo.fa")
        p = subprocess.run([BIN, "maskopt", "-t", alg, "-k", "31", "-o", mo, ms], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
        assert p.returncode == 0 and open(mo).read() == ref.stdout, alg
        assert "result in the mapping" in p.stderr and "result written" in p.stderr, p.stderr[-600:]   # (the streamed path did run, all the way)
    # two records: the reference optimises the first and then objects (masks.h:258); the streamed path hands such files to the host parser
    with open(ms, "a") as f:
        f.write(">second\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n")
    a = subprocess.run([BIN, "maskopt", "-k", "31", ms], capture_output=True, text=True)
    b = subprocess.run([BIN, "maskopt", "-k", "31", ms], capture_output=True, text=True, env=dict(os.environ, KC_CLI_NOSTREAM="1"))
    assert a.returncode == b.returncode and a.stdout == b.stdout and a.returncode != 0


@pytest.mark.gpu
def test_cli_gpus_option_is_byte_identical(tmp_path, golden_dir):
    """`compute --gpus N` (new long option / KMERCAMEL_GPUS): N rank threads in one process, the k-mer set
    hash-sharded over them, exchanges through the in-process thread transport (ranks beyond the device count share the GPU).
    Same bytes as without the option: random reads, branching reads, a genome with cycles (one record: some ranks get no
    input); `-M` and `-z` on the sharded path; what it does not cover runs on one GPU and says so."""
    import gzip
    recs = gen.random_records(3000, 400, seed=21)
    fa = str(tmp_path / "reads.fa")
    with open(fa, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))
    br = str(tmp_path / "branching.fa")
    with open(br, "wb") as f:
        for i, r in enumerate(gen.stress_reads(4000, 150, 250, seed=77)):
            f.write(b">b%d\n%s\n" % (i, r))
    genome = str(tmp_path / "sp.fa")
    with gzip.open(os.path.join(golden_dir, "spneumoniae.fa.gz"), "rb") as g, open(genome, "wb") as f:
        f.write(g.read())
    for path, ks in ((fa, ("31", "21")), (br, ("23",)), (genome, ("31",))):
        for k in ks:
            for extra in ((), ("-u",)):
                want = run("compute", "-k", k, *extra, path)
                assert want.returncode == 0, want.stderr[-400:]
                for n in (("2", "3") if path != genome else ("2",)):
                    got = run("compute", "-k", k, *extra, "--gpus", n, path)
                    assert got.returncode == 0, got.stderr[-600:]
                    assert got.stdout == want.stdout, (path, k, extra, n)
    # through the environment, into a file
    o1, o2 = str(tmp_path / "a.fa"), str(tmp_path / "b.fa")
    assert run("compute", "-k", "31", "-o", o1, fa).returncode == 0
    p = subprocess.run([BIN, "compute", "-k", "31", "-o", o2, fa], capture_output=True, text=True, env=dict(os.environ, KMERCAMEL_GPUS="2"))
    assert p.returncode == 0 and open(o1).read() == open(o2).read()
    # -M on the sharded path: the mask probes go to the owners of the k-mers
    for path, k in ((fa, "31"), (br, "23"), (genome, "31")):
        m1, m2 = str(tmp_path / "m1.fa"), str(tmp_path / "m2.fa")
        a = run("compute", "-k", k, "-M", m1, path)
        b = run("compute", "-k", k, "-M", m2, "--gpus", "3", path)
        assert a.returncode == 0 and b.returncode == 0, b.stderr[-500:]
        assert a.stdout == b.stdout and open(m1).read() == open(m2).read(), (path, k)
    # -z on the sharded path: the occurrences of a k-mer are counted where they all meet, at its owner (overlapping reads: k-mers
    # that occur twice or three times exist)
    for z in ("2", "3"):
        for extra in ((), ("-u",)):
            want = run("compute", "-k", "23", "-z", z, *extra, br)
            got = run("compute", "-k", "23", "-z", z, *extra, "--gpus", "3", br)
            assert want.returncode == 0 and got.returncode == 0, got.stderr[-500:]
            assert "running on one GPU" not in got.stderr and got.stdout == want.stdout, (z, extra)
    # -S on the sharded path: the records are the simplitigs (record order), no k-mer table involved (any k)
    sf = str(tmp_path / "simplitigs.fa")
    with open(sf, "wb") as f:
        for i, r in enumerate(gen.random_records(900, 77, seed=9) + gen.random_records(50, 31, seed=10)):
            f.write(b">s%d\n%s\n" % (i, r))
    for k in ("31", "15"):
        for extra in ((), ("-u",)):
            want = run("compute", "-k", k, "-S", *extra, sf)
            got = run("compute", "-k", k, "-S", *extra, "--gpus", "3", sf)
            assert want.returncode == 0 and got.returncode == 0, got.stderr[-500:]
            assert "running on one GPU" not in got.stderr and got.stdout == want.stdout, (k, extra)
    # small k: one GPU, with a note
    p = run("compute", "-k", "15", "--gpus", "2", br)
    assert p.returncode == 0 and "running on one GPU" in p.stderr and p.stdout == run("compute", "-k", "15", br).stdout
    # an input whose simplitigs are too many for the dense overlap graph (S >= N / 5, main.cpp:175): the ranks find out together
    # after phase 1 and the CLI runs the sparse branch on one GPU
    sp = str(tmp_path / "sparse.fa")
    with open(sp, "wb") as f:
        for i, r in enumerate(gen.random_records(4000, 33, seed=5)):   # 3 k-mers per record at k = 31
            f.write(b">s%d\n%s\n" % (i, r))
    p = run("compute", "-k", "31", "--gpus", "2", sp)
    # (round 6: the library keeps such a request inside the collective -- rank 0 runs it and every rank gets its slice -- so the CLI's own
    # one-GPU fallback with its note is only taken when the library hands the request back, KC_SHARD_NO_ROOT_FALLBACK=1)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout == run("compute", "-k", "31", sp).stdout
    p = subprocess.run([BIN, "compute", "-k", "31", "--gpus", "2", sp], capture_output=True, text=True, env=dict(os.environ, KC_SHARD_NO_ROOT_FALLBACK="1"))
    assert p.returncode == 0 and "sparse branch" in p.stderr and "running on one GPU" in p.stderr, p.stderr[-500:]
    assert p.stdout == run("compute", "-k", "31", sp).stdout
    # the same through stdin: the ranks have consumed the stream, the one-GPU run must work from what they read (a second read of `-`,
    # a FIFO or a process substitution finds nothing)
    with open(sp, "rb") as f:
        q = subprocess.run([BIN, "compute", "-k", "31", "--gpus", "2", "-"], stdin=f, capture_output=True)
    assert q.returncode == 0, q.stderr[-500:]
    assert q.stdout.split(b"\n", 1)[1] == p.stdout.encode().split(b"\n", 1)[1]   # (the header names the path)
    assert run("compute", "-k", "31", "--gpus", "0", fa).returncode == 1
    # KMERCAMEL_GPUS: a value that is no number between 1 and 64 is ignored with a warning (not "--gpus must be ..."), and the
    # variable never prints the one-GPU note for commands the sharded path does not cover
    for bad in ("", "two", "0", "999"):
        e = subprocess.run([BIN, "compute", "-k", "31", fa], capture_output=True, text=True, env=dict(os.environ, KMERCAMEL_GPUS=bad))
        assert e.returncode == 0 and "KMERCAMEL_GPUS" in e.stderr and e.stdout == run("compute", "-k", "31", fa).stdout, e.stderr[-300:]
    e = subprocess.run([BIN, "compute", "-k", "15", br], capture_output=True, text=True, env=dict(os.environ, KMERCAMEL_GPUS="2"))
    assert e.returncode == 0 and "running on one GPU" not in e.stderr
