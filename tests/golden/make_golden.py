#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json.  Runs only where /root/reference exists: the compute goldens come from the
reference's own headers compiled in place (oracle/_ref, see oracle/Makefile + oracle/ref_shim.cpp); the maskopt goldens
come from the oracle restatement of masks.h:40-78 (masks.h itself needs GLPK and is not buildable here) after it has been
pinned on the same data: oracle max-one(reference -o line) == reference -M line.

The data files next to this script (spneumoniae.fa.gz, simplitigs-k31.fa.gz, global-k*c.fa.gz, *.fa) are byte-identical
(gzip'ed) copies of the DATA files the reference's tests use (data/, tests/testdata/)."""
import gzip
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402


def md5(s):
    return hashlib.md5(s.encode()).hexdigest()


def main():
    ref = oracle_lib.Reference()
    orc = oracle_lib.Oracle()
    out = {"generator": "tests/golden/make_golden.py against oracle/_ref (reference v2.3.0 headers compiled in place)"}
    tmp = tempfile.mkdtemp()

    def unz(name):
        p = os.path.join(tmp, name[:-3])
        with gzip.open(os.path.join(HERE, name), "rb") as f, open(p, "wb") as g:
            g.write(f.read())
        return p

    sp = unz("spneumoniae.fa.gz")
    st_path = unz("simplitigs-k31.fa.gz")
    rc, o, m, st = ref.compute_file(st_path, 31, True, True, want_mask=True, tmpdir=tmp)
    assert rc == 0
    out["simplitigs-k31"] = {"N": st[0], "S": st[1], "L": st[2], "md5": md5(o), "md5_mask": md5(m)}
    out["spneumoniae"] = {}
    for k in (13, 31, 63, 127):
        for c in (1, 0):
            rc, o, m, st = ref.compute_file(sp, k, bool(c), False, want_mask=True, tmpdir=tmp)
            assert rc == 0
            kset, ones = oracle_lib.kmers_of_masked_superstring(o, k, bool(c))
            assert ones == st[0] == len(kset)
            # pin of the oracle's max-one restatement on reference data
            rc2, mo = orc.maskopt(o, k, bool(c), False)
            assert rc2 == 0 and mo == m, "oracle max-one != reference -M"
            out["spneumoniae"]["%d_%d" % (k, c)] = {"N": st[0], "S": st[1], "L": st[2], "md5": md5(o), "md5_mask": md5(m),
                                                    "set_md5": md5("".join(sorted(kset)))}
            print(k, c, out["spneumoniae"]["%d_%d" % (k, c)], flush=True)
    out["maskopt"] = {}
    for k in (13, 31, 63, 127):
        p = unz("global-k%dc.fa.gz" % k)
        recs = orc.parse_records(p)
        ms = recs[0][2]
        d = {}
        for alg, mini in (("max-one", False), ("min-one", True)):
            rc, r = orc.maskopt(ms, k, True, mini)
            assert rc == 0
            d[alg] = md5(r)
        kset, _ = oracle_lib.kmers_of_masked_superstring(ms, k, True)
        d["N"] = len(kset)
        assert d["N"] == out["spneumoniae"]["%d_1" % k]["N"]
        out["maskopt"][str(k)] = d
    # format converters: the reference's conversions.h compiled in place, on the reference's own test data
    out["converters"] = {}
    for k in (13, 31, 63, 127):
        p = unz("global-k%dc.fa.gz" % k)
        sup, mask = ref.split_ms(p)
        spss = ref.ms_to_spss(p, k)
        sp_path = os.path.join(tmp, "spss-k%d.fa" % k)
        with open(sp_path, "w") as f:
            f.write(spss)
        back = ref.spss_to_ms(sp_path, k)
        assert back.startswith(">superstring " + sp_path + "\n")
        out["converters"][str(k)] = {"superstring": md5(sup), "mask": md5(mask), "joined": md5(ref.join_ms(sup, mask)), "spss": md5(spss),
                                     "spss_ms_seq": md5(back.split("\n")[1])}
    out["converters"]["spss.fa"] = {str(k): md5(ref.spss_to_ms(os.path.join(HERE, "spss.fa"), k).split("\n", 1)[1]) for k in (3, 5)}
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
