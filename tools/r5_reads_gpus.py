"""round 5: reads with coverage and errors (branch ports everywhere) through `compute --gpus 2`, rank threads sharing the GPU: where the sharded
flow's time goes on branching input.  python tools/r5_reads_gpus.py [genome_Mbp] [coverage]"""
import os, subprocess, sys, time, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gm = float(sys.argv[1]) if len(sys.argv) > 1 else 50
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(7)
G, rl, er = int(gm * 1e6), 150, 0.01
n_reads = int(G * cov / rl)
genome = rng.integers(0, 4, size=G, dtype=np.uint8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
d = "/dev/shm/kc_r5rg"; os.makedirs(d, exist_ok=True)
hdr = np.frombuffer(b">read_0000000\n", dtype=np.uint8)
with open(d + "/reads.fa", "wb") as f:
    for lo in range(0, n_reads, 500000):
        hi = min(n_reads, lo + 500000)
        pos = rng.integers(0, G - rl, size=hi - lo)
        codes = genome[pos[:, None] + np.arange(rl)[None, :]]
        rev = rng.random(hi - lo) < 0.5
        codes[rev] = 3 - codes[rev][:, ::-1]
        err = rng.random(codes.shape) < er
        codes[err] = (codes[err] + rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)) & 3
        a = np.empty((hi - lo, len(hdr) + rl + 1), dtype=np.uint8)
        a[:, :len(hdr)] = hdr; a[:, len(hdr):len(hdr) + rl] = acgt[codes]; a[:, -1] = 10
        f.write(a.tobytes())
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
ref = None
for gp in ((None, "2", "4") if "--trace" not in sys.argv else ("2",)):
    out = d + "/o.txt"
    if os.path.exists(out): os.unlink(out)
    time.sleep(3)
    t0 = time.time()
    p = subprocess.run([exe, "compute", "-k", "31"] + (["--gpus", gp] if gp else []) + ["-o", out, d + "/reads.fa"], capture_output=True, text=True, env=dict(os.environ, KC_SHARD_TRACE="1"))
    dt = time.time() - t0
    m = hashlib.md5(open(out, "rb").read()).hexdigest() if p.returncode == 0 else None
    if ref is None: ref = m
    steps = [l.split("step ")[1][:46].strip() for l in p.stderr.splitlines() if "[shard 0/" in l and " step " in l]
    info = [l for l in p.stderr.splitlines() if "branch edges" in l and "[shard 0/" in l]
    print("gpus %s: wall %.2f s rc %d identical %s | %s | %s" % (gp, dt, p.returncode, m == ref, " | ".join(steps), info[0][-90:] if info else ""), flush=True)
    if "--trace" in sys.argv:
        print("\n".join(l[:200] for l in p.stderr.splitlines() if "[shard 0/" in l or "[simp]" in l))
import shutil; shutil.rmtree(d)
