import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, kmercamel_amd as kc
ctx = kc.Context(0)
seq = bench.generate(1, 0, 100000, 1030)
for it in range(3):
    r = kc.compute(ctx, seq, 31, True)
    r.close()
