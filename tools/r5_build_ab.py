"""round 5: what bounds the set build?  One process, cfg[2]-shaped input resident on the device, the build repeated under experiment knobs;
per-kernel times come from rocprofv3 --kernel-trace --stats around this script.  python tools/r5_build_ab.py [records] [VAR=v,VAR=v ...]..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import kmercamel_amd as kc
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
variants = sys.argv[2:] or [""]
rng = np.random.default_rng(2)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
L = 1030
parts = []
for c in range(0, nrec, 100000):
    n = min(100000, nrec - c)
    a = acgt[rng.integers(0, 4, size=(n, L + 1), dtype=np.uint8)]
    a[:, L] = 10
    parts.append(a.reshape(-1))
seq = np.concatenate(parts)[:-1]
del parts
ctx = kc.Context(0)
d = torch.from_numpy(seq).cuda()
k = 31
for v in variants:
    env = dict(x.split("=") for x in v.split(",") if x)
    for a, b in env.items(): os.environ[a] = b
    ts = []
    size = None
    for i in range(int(os.environ.get("AB_REPS", "3"))):
        if os.environ.get("AB_SLEEP"): time.sleep(float(os.environ["AB_SLEEP"]))   # (an idle device runs the scatter faster: see DESIGN.md)
        s = kc.KmerSet(ctx, k, True, capacity_hint=len(seq))
        torch.cuda.synchronize()
        t0 = time.time()
        try:
            s.add_device(d.data_ptr(), len(seq))
            size = s.size()
        except Exception as e:
            size = "error: %s" % str(e)[:80]
        ts.append((time.time() - t0) * 1e3)
        s.close()
    for a in env: del os.environ[a]
    print("%-40s build wall ms %s size %s" % (v or "(default)", " ".join("%.1f" % t for t in ts), size), flush=True)
ctx.close()
