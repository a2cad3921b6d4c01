"""CPU tests (not gpu): the oracle against the golden vectors and the reference's unit KATs, the host logic, the C-ABI
library surface, and the phase-2 bucket logic through the CPU simulation that shares kc_overlap_logic.h with the kernels."""
import ctypes
import gzip
import hashlib
import json
import os
import random
import re
import subprocess

import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NONE = (1 << 64) - 1


def _golden():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


# ---------------------------------------------------------------------------------- A1/A4/A5 primitives (kmers_unittest.h)
def test_kmer_primitive_kats(oracle):
    o = oracle
    big = 0b11111101111111111111111111111111111111111111111111111111111110
    assert o.bit_suffix(0b1100011110, 5, 3) == 0b011110 and o.bit_suffix(0b1110, 2, 2) == 0b1110 and o.bit_suffix(0b1110, 2, 0) == 0
    assert o.bit_suffix(big, 31, 23) == 0b1111111111111111111111111111111111111111111110 and o.bit_suffix(big, 31, 7) == 0b11111111111110
    assert o.bit_prefix(0b1100011110, 5, 3) == 0b110001 and o.bit_prefix(0b1110, 2, 2) == 0b1110 and o.bit_prefix(0b1110, 2, 0) == 0
    assert o.bit_prefix(big, 31, 23) == 0b1111110111111111111111111111111111111111111111 and o.bit_prefix(big, 31, 7) == 0b11111101111111
    # ReverseComplement (kmers_unittest.h:94-112)
    assert o.reverse_complement(0b1001, 2) == 0b1001 and o.reverse_complement(0b101111, 3) == 0b000001 and o.reverse_complement(0b11, 1) == 0
    assert o.reverse_complement(big, 31) == 0b01000000000000000000000000000000000000000000000000000010000000
    assert [o.nucleotide_to_int(c) for c in "ACGTacgtNB\n"] == [0, 1, 2, 3, 0, 1, 2, 3, 4, 4, 4]
    assert o.masked("a", 1) == "A" and o.masked("A", 0) == "a" and o.masked("T", 1) == "T" and o.masked("g", 0) == "g"
    rng = random.Random(3)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    for k in (1, 2, 31, 32, 63, 64, 127):
        for _ in range(20):
            s = "".join(rng.choice("ACGT") for _ in range(k))
            rc = "".join(comp[c] for c in reversed(s))
            assert o.reverse_complement(oracle_lib.encode_kmer(s), k) == oracle_lib.encode_kmer(rc)


def test_parser_kats(oracle):
    # tests/parser_unittest.h:38-63
    path = os.path.join(GOLDEN, "test.fa")
    for k, c, cs, want in [(10, True, False, 3), (10, True, True, 2), (5, True, False, 11), (5, False, False, 11), (2, True, False, 9),
                           (2, False, False, 11)]:
        assert oracle.read_kmers_count(path, k, c, cs) == want
    # tests/parser_unittest.h:19-36 (records of test.fa)
    recs = oracle.parse_records(path)
    assert [(r[0], r[2]) for r in recs] == [("1", "ACCCGAAC"), ("2", "CGTANATGC"), ("3", "AcCCGTTTAACG"), ("4", "A")]


def test_host_record_reader_matches_oracle(oracle, tmp_path):
    import kmercamel_amd as kc
    cases = {
        "plain.fa": b">a desc x\nACGT\nAC\n>b\n\nGG\n",
        "crlf.fa": b">a\r\nACGT\r\nAC\r\n>b\r\nGG\r\n",
        "noeol.fa": b">a\nACGT",
        "lead.fa": b"garbage\n>a\nAC\n",
        "fq.fq": b"@r1\nACGT\n+\nIIII\n@r2 c\nGGA\n+r2\nII>\n",
        "empty.fa": b"",
        "hdr_only.fa": b">",
        "empty_rec.fa": b">a\n>b\nAC\n",
    }
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        want = oracle.parse_records(str(p))
        got = kc.read_records(str(p))
        assert [(a, b, c.decode()) for a, b, c in got] == [tuple(w) + ("",) * (3 - len(w)) for w in want], name
    gz = tmp_path / "x.fa.gz"
    gz.write_bytes(gzip.compress(cases["plain.fa"]))
    assert [(a, c.decode()) for a, _, c in kc.read_records(str(gz))] == [("a", "ACGTAC"), ("b", "GG")]
    assert [(r[0], r[2]) for r in oracle.parse_records(str(gz))] == [("a", "ACGTAC"), ("b", "GG")]


# ---------------------------------------------------------------------------------- B/C KATs against the oracle
def test_simplitig_kats(oracle):
    enc = oracle_lib.encode_kmer
    # tests/simplitigs_unittest.h:183-208 (order tolerant: sorted by length), both simplitig orders
    for mode in (0, 1):
        for kmers, k, want in [(["GCT", "TAA", "AAA"], 3, ["GCT", "TAAA"]), (["TAA", "AAA", "GCT"], 3, ["GCT", "TAAA"]),
                               (["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTG", "TTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"], 31, ["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"])]:
            got = oracle.simplitigs_of_kmers([enc(x) for x in kmers], k, False, mode)
            assert sorted(got, key=len) == want
    # tests/simplitigs_unittest.h:142-181 NextSimplitig (accepts the reverse complement too)
    assert oracle.simplitigs_of_kmers([0b00010000, 0b00000100], 4, False, 0)[0] == "AACAA"
    got = oracle.simplitigs_of_kmers([0b00010000, 0b00000011, 0b00000100], 4, True, 0)
    assert any(x in ("AACAA", "TTGTT") for x in got) and sorted(len(x) for x in got) == [4, 5]


OHP_KATS = [  # tests/global_unittest.h:53-111
    (["AT"], 2, True, False, [NONE, NONE], [255, 255]),
    (["ACG", "TAC", "GGC"], 3, False, False, [2, 0, NONE], [1, 2, 255]),
    (["ACAA", "ATTT", "AACA"], 4, True, False, [4, 3, 0, 5, NONE, NONE], [2, 2, 3, 3, 255, 255]),
    (["ACG", "CGT", "TAA"], 3, False, True, [1, 2, 0], [2, 1, 1]),
    (["ACGT", "TAA"], 3, False, True, [1, 0], [1, 1]),
    (["ACC", "CGG"], 3, True, True, [3, 2, 0, 1], [2, 2, 0, 2]),
]


def test_overlap_path_kats(oracle):
    for strs, k, c, lb, ef, ov in OHP_KATS:
        assert oracle.overlap_path(strs, k, c, lb) == (ef, ov)


def test_superstring_and_global_kats(oracle):
    # tests/global_unittest.h:11-51
    for ef, ov, strs, k, want, c in [([2, 0, NONE], [1, 2, 255], ["ACG", "TAC", "GGC"], 3, "TAcGgc", False),
                                     ([4, 3, 1, NONE, 5, NONE], [1, 2, 1, 255, 2, 255], ["GCC", "ACG", "TAC"], 3, "GcCGta", True),
                                     ([3, 1, NONE, NONE], [1, 1, 255, 255], ["GCC", "TACG"], 3, "GcCGta", True)]:
        out, _, L = oracle.superstring_from_path(strs, ef, ov, k, c)
        assert out == want and L == len(want)
    # tests/global_unittest.h:113-138
    for want, k, strs, c in [("TACgt", 3, ["CGT", "TAC", "ACG"], False), ("ACGT", 1, ["ACGT"], False), ("ACgTtt", 3, ["CGT", "TTT", "ACG"], False),
                             ("ACgTtt", 3, ["ACGT", "TTT"], False), ("TActt", 4, ["TACT", "ACTT"], False),
                             ("TActTaaGgac", 4, ["TACT", "ACTT", "GGAC", "TAAG"], False),
                             ("TTtcttttttttttttttttttttttttttga", 31, ["TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTG", "TTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA"], False),
                             ("AtTTgtt", 4, ["ACAA", "ATTT", "AACA"], True)]:
        assert oracle.global_(strs, k, c)[1] == want
    assert oracle.global_([], 3, True)[0] == -2   # "input cannot be empty" (global.h:219-221)


def test_lower_bound_kats(oracle):
    # tests/lower_bound_unittest.h:11-41
    for strs, k, c, want in [(["ACG", "CGT", "TAA"], 3, False, 5), (["ACGT", "TAA"], 3, False, 5), (["AC", "GG"], 2, True, 3),
                             (["AAACCC", "CCCAAA", "TGGGGT"], 6, False, 11)]:
        assert oracle.lower_bound(strs, k, c) == want


def test_maskopt_kats(oracle):
    # tests/masks_unittest.h:19-65 on tests/testdata/masktest.fa ("ACGTAACGT"): the k-mer set of each case is the set that
    # the expected min-one line itself represents, so Optimize() (masks.h:239-261) applied to that line reproduces the case.
    for ms, k, c, mini, want in [("ACgTaacgt", 3, False, True, "ACgTaacgt"), ("ACgTaacgt", 3, False, False, "ACgTaACgt"),
                                 ("AcgTaacgt", 3, True, True, "AcgTaacgt"), ("AcgTaacgt", 3, True, False, "ACgTaACgt")]:
        assert oracle.maskopt(ms, k, c, mini) == (0, want)
    assert oracle.maskopt("ACGNACG", 3, True, False)[0] == -3


# ---------------------------------------------------------------------------------- goldens from the reference
def test_oracle_reproduces_S_golden(oracle, tmp_path):
    g = _golden()["simplitigs-k31"]
    rc, out, mask, st = oracle.compute_file(os.path.join(GOLDEN, "simplitigs-k31.fa.gz"), 31, True, True, mode=0)
    assert rc == 0 and (st[0], st[1], st[2]) == (g["N"], g["S"], g["L"])
    assert md5(out) == g["md5"] and md5(mask) == g["md5_mask"]


@pytest.mark.parametrize("k,c", [(31, 1), (13, 1), (63, 0), (127, 1)])
def test_oracle_reference_order_reproduces_compute_goldens(oracle, k, c):
    # byte-identical to the compiled reference (khash bucket-order emulation); remaining (k, mode) pairs are covered by
    # tests/test_oracle_vs_reference.py where /root/reference exists
    g = _golden()["spneumoniae"]["%d_%d" % (k, c)]
    rc, out, mask, st = oracle.compute_file(os.path.join(GOLDEN, "spneumoniae.fa.gz"), k, bool(c), False, mode=0)
    assert rc == 0 and (st[0], st[1], st[2]) == (g["N"], g["S"], g["L"])
    assert md5(out) == g["md5"] and md5(mask) == g["md5_mask"]


def test_oracle_canonical_order_is_valid(oracle):
    # the GPU's rule restated on the CPU: same k-mer set, min-one mask, -M consistent, L close to the reference's
    g = _golden()["spneumoniae"]["31_1"]
    rc, out, mask, st = oracle.compute_file(os.path.join(GOLDEN, "spneumoniae.fa.gz"), 31, True, False, mode=1)
    assert rc == 0 and st[0] == g["N"]
    kset, ones = oracle_lib.kmers_of_masked_superstring(out, 31, True)
    assert ones == g["N"] and md5("".join(sorted(kset))) == g["set_md5"]
    assert oracle.maskopt(out, 31, True, False) == (0, mask)
    assert abs(st[2] - g["L"]) / g["L"] < 1e-3


def test_maskopt_fixture_goldens(oracle):
    g = _golden()["maskopt"]
    for k in (31, 127):
        ms = oracle.parse_records(os.path.join(GOLDEN, "global-k%dc.fa.gz" % k))[0][2]
        for alg, mini in (("max-one", False), ("min-one", True)):
            rc, out = oracle.maskopt(ms, k, True, mini)
            assert rc == 0 and md5(out) == g[str(k)][alg]
        kset, _ = oracle_lib.kmers_of_masked_superstring(oracle.maskopt(ms, k, True, True)[1], k, True)
        assert len(kset) == g[str(k)]["N"]


# ---------------------------------------------------------------------------------- C ABI surface (no compute without a GPU)
def test_abi_library_exports_every_declared_symbol():
    import kmercamel_amd as kc
    header = open(os.path.join(ROOT, "include", "kmercamel_amd.h")).read()
    declared = set(re.findall(r"\b(kc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(kc.api.SIGNATURES), declared ^ set(kc.api.SIGNATURES)
    lib = kc.api.load()
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", kc.api.LIB_PATH], text=True)
    exported = set(re.findall(r" T (kc_[a-z0-9_]+)", out))
    assert declared <= exported
    assert b"gfx950" in lib.kc_version()


def test_abi_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import kmercamel_amd as kc
    with pytest.raises(kc.KcError) as e:
        kc.Context(0)
    assert e.value.code == kc.api.KC_E_NO_DEVICE


def test_launch_guard_refuses_2pow32_threads_at_the_call_site():
    """kc_grid / KC_LAUNCH / KC_HIP (csrc/kc_internal.h): a 1-D launch of 2^32 or more threads returns KC_E_ARG from the calling function
    before anything is launched, and the overflow flag never goes stale (tests/sim/grid_check.hip, host only: nothing reaches a GPU)."""
    exe = os.path.join(ROOT, "tests", "sim", "grid_check")
    src = os.path.join(ROOT, "tests", "sim", "grid_check.hip")
    hdr = os.path.join(ROOT, "kmercamel_amd", "csrc", "kc_internal.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["hipcc", "-std=c++17", "-O1", "--offload-arch=gfx950", "-o", exe, src, "-L" + os.path.join(ROOT, "kmercamel_amd"),
                               "-lkmercamel_amd", "-Wl,-rpath,$ORIGIN/../../kmercamel_amd"], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "grid_check ok" in out.stdout, out.stdout + out.stderr
    # and no kernel of the library is launched any other way
    for f in os.listdir(os.path.join(ROOT, "kmercamel_amd", "csrc")):
        if f.endswith((".hip", ".h")) and f != "kc_internal.h":
            txt = open(os.path.join(ROOT, "kmercamel_amd", "csrc", f)).read()
            assert "hipLaunchKernelGGL" not in txt and "<<<" not in txt, f


def test_shipped_library_has_no_wrong_result_knobs():
    """Timing experiments that skip stores / inserts (KC_SCATTER_EXP, KC_BUILD_EXP: the results are wrong then) are compiled only into
    the tools' build (-DKC_EXPERIMENTS, `make experiments`); the shipped library does not even contain their names."""
    lib = os.path.join(ROOT, "kmercamel_amd", "libkmercamel_amd.so")
    data = open(lib, "rb").read()
    assert b"_EXP" not in data.replace(b"KC_EXPERIMENTS", b""), "an experiment knob is compiled into the release library"
    # and the sources read them through KC_EXP_KNOB only
    for f in os.listdir(os.path.join(ROOT, "kmercamel_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(ROOT, "kmercamel_amd", "csrc", f)).read()
            assert not re.search(r'getenv\("KC_[A-Z_]*_EXP"\)', txt), f


def test_probe_kernel_variants_do_not_spill_vector_registers(tmp_path):
    """Every instantiation of the dominant kernel (kc_probe_block_kernel: single / double / quadruple-word keys, compile-time and run-time k,
    the sharded form) must fit its register budget without spilling VECTOR registers to scratch memory: the kernel sits on that cliff --
    in round 5 a one-instruction-per-position saving in the minimizer pass took the 127-mer form from 123 registers and no spill to
    128 + 98 spilled, 7.6 -> 22 ms per 10^8 k-mers, and nothing noticed (no GPU needed for this check: hipcc cross-compiles)."""
    import subprocess
    out = []
    for src in ("kc_simplitigs.hip", "kc_shard.hip"):
        asm = str(tmp_path / (src + ".s"))
        subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-result", "--cuda-device-only", "-S", "-o", asm,
                               os.path.join(ROOT, "kmercamel_amd", "csrc", src)], stderr=subprocess.DEVNULL)
        name = None
        for line in open(asm):
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split()[1]
            elif line.startswith(".vgpr_spill_count:") and name and "kc_probe_block_kernel" in name:
                out.append((name, int(line.split()[1])))
    assert len(out) >= 20, "the probe kernel's instantiations were not found in the assembly"
    bad = [(n, c) for n, c in out if c != 0]
    assert not bad, "probe kernel variants spill vector registers: %s" % bad


def test_product_never_touches_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "kmercamel_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                # (comments may cite oracle/kc_oracle.cpp as the restatement of a shared spec; loading/linking/including it is what is banned)
                assert "oracle_lib" not in txt and "libkcoracle" not in txt and "libkcref" not in txt and "oracle/_ref" not in txt, f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', txt), f
                # the only library the product loads at run time is RCCL (kc_comm.hip, multi-GPU transport)
                if "dlopen" in txt:
                    assert f == "kc_comm.hip" and "oracle" not in txt.lower() and all("rccl" in so for so in re.findall(r'"([^"]*\.so[^"]*)"', txt)), f


# ---------------------------------------------------------------------------------- phase-2 bucket logic via the CPU simulation
def _sim():
    so = os.path.join(ROOT, "tests", "sim", "libkcsim.so")
    src = os.path.join(ROOT, "tests", "sim", "sim_overlap.cpp")
    hdr = os.path.join(ROOT, "kmercamel_amd", "csrc", "kc_overlap_logic.h")
    hdr2 = os.path.join(ROOT, "kmercamel_amd", "csrc", "kc_overlap_lds.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(hdr2)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


def _sim_path(lib, strs, k, c, lb, seed, lds=False):
    n = len(strs)
    V = n * (2 if c else 1)
    arr = (ctypes.c_char_p * n)(*[s.encode() for s in strs])
    ef = (ctypes.c_uint64 * V)()
    ov = (ctypes.c_uint8 * V)()
    rep = ctypes.c_uint64(0)
    rc = lib.sim_overlap_path2(arr, ctypes.c_uint64(n), k, int(c), int(lb), ef, ov, ctypes.c_uint64(seed), ctypes.byref(rep), int(lds))
    return rc, list(ef), list(ov), rep.value


def test_bucket_logic_simulation_matches_oracle(oracle):
    lib = _sim()
    for strs, k, c, lb, ef, ov in OHP_KATS:
        assert _sim_path(lib, strs, k, c, lb, 1)[:3] == (0, ef, ov)
        assert _sim_path(lib, strs, k, c, lb, 1, lds=True)[:3] == (0, ef, ov)
    rng = random.Random(12)
    repairs = 0
    for it in range(120):
        k = rng.choice([2, 3, 4, 5, 7, 9, 13, 31, 33, 63, 70, 127]) if it % 3 else rng.choice([3, 4, 5])
        n = rng.choice([1, 2, 3, 5, 10, 40, 200, 600])
        alpha = "ACGT" if rng.random() < 0.7 else "AC"
        base = "".join(rng.choice(alpha) for _ in range(max(k, 400)))
        strs = []
        for _ in range(n):
            L = k + rng.choice([0, 0, 1, 2, 5, 20])
            if rng.random() < 0.5:
                p = rng.randrange(0, len(base) - L + 1)
                strs.append(base[p:p + L])
            else:
                strs.append("".join(rng.choice(alpha) for _ in range(L)))
        for c in (True, False):
            for lb in (False, True):
                want = oracle.overlap_path(strs, k, c, lb)
                rc, ef, ov, rep = _sim_path(lib, strs, k, c, lb, it + 7)
                assert rc == 0 and (ef, ov) == want, (k, n, c, lb)
                repairs += rep
                # the same through the replay the device runs out of LDS (32-bit ids, linked event lists, one body per bucket kind)
                rc, ef, ov, rep = _sim_path(lib, strs, k, c, lb, it + 7, lds=True)
                assert rc == 0 and (ef, ov) == want, ("lds replay", k, n, c, lb)
    assert repairs > 0   # the verify/repair loop was exercised


def test_real_data_shape_generators_are_deterministic_and_shaped():
    """tools/shapes.py feeds the full-scale runs and the GPU tests: same seed, same bytes; records of the stated length over ACGT;
    the coverage reads carry both strands and substitution errors (the reads' k-mers are a superset of most of the genome's)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import shapes
    g1, cfg = shapes.make_input("genome", 0.0004)
    g2, _ = shapes.make_input("genome", 0.0004)
    assert np.array_equal(g1, g2) and cfg["records"] == 25 and cfg["record_bp"] == 40000
    recs = bytes(g1).split(b"\n")
    assert len(recs) == 25 and all(len(r) == 40000 and set(r) <= set(b"ACGT") for r in recs)
    c1, ccfg = shapes.make_input("coverage", 0.001)
    c2, _ = shapes.make_input("coverage", 0.001)
    assert np.array_equal(c1, c2)
    reads = bytes(c1).split(b"\n")
    assert len(reads) == int(ccfg["genome_bp"] * ccfg["coverage"] / ccfg["read_bp"]) and all(len(r) == 150 for r in reads)
    # tandem repeats are really there: some record holds a unit repeated back to back
    recs2 = shapes.genome_records(2, 40000, 5, repeats_per_mbp=0.0, tandems_per_mbp=200.0)
    s = bytes(shapes.ACGT[recs2[0]])
    assert any(s[i:i + 60] == s[i + u:i + u + 60] for u in range(40, 400) for i in range(0, len(s) - 500, 997))
