import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, kmercamel_amd as kc
ctx = kc.Context(0)
for recs in (2000, 100000):
    seq = bench.generate(1, 0, recs, 1030)
    for env in ({}, {"KC_NOFRAG": "1"}):
        for k_, v_ in env.items(): os.environ[k_] = v_
        for it in range(2):
            r = kc.compute(ctx, seq, 31, True)
        print(recs, env, {k: r.stats[k] for k in ("n_kmers", "n_simplitigs", "us_simplitig_phase", "us_total")}, flush=True)
        r.close()
        for k_ in env: del os.environ[k_]
