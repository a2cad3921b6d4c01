#!/usr/bin/env python3
"""one -S case through the library: python tools/ovg_case.py <fasta> <k> (debugging aid for the grouped overlap levels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmercamel_amd as kc
import oracle_lib
fa, k = sys.argv[1], int(sys.argv[2])
ctx = kc.Context(0)
r = kc.compute_file(ctx, fa, k, True, want_mask=True, assume_simplitigs=True)
out, mask = r.strings(want_mask=True)
rc, want, wmask, st = oracle_lib.Oracle().compute_file(fa, k, True, True, mode=0)
print("ok" if (rc == 0 and out == want and mask == wmask) else "MISMATCH", len(out), r.stats.get("overlap_repairs"))
