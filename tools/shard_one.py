#!/usr/bin/env python3
"""One rank of the hash-sharded flow on its own (self transport): `python tools/shard_one.py [records] [steps]`.  Used under
rocprofv3 to see what the sharded kernels cost next to the single-GPU flow on the same records (no exchange involved)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import kmercamel_amd as kc

records = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seq = bench.generate(2, 0, records, 1030)
d = torch.from_numpy(seq).cuda()
ctx = kc.Context(0)
comm = kc.Comm.self_(ctx)
for mode in ("sharded", "single"):
    for i in range(steps + 1):
        torch.cuda.synchronize(); t0 = time.time()
        if mode == "sharded": r = kc.compute_sharded(ctx, comm, None, 31, True, device_ptr=d.data_ptr(), length=d.numel())
        else: r = kc.compute(ctx, None, 31, True, device_ptr=d.data_ptr(), length=d.numel())
        torch.cuda.synchronize(); dt = time.time() - t0
        n = r.stats["n_kmers"]; r.close()
        if i: print("%s: %.1f ms for %d k-mers" % (mode, dt * 1e3, n), flush=True)
comm.close(); ctx.close()
