"""round 5: the genome shape (long chains, repeats, cycles) through `compute --gpus N`, rank threads sharing the one GPU, against the plain CLI:
identical bytes, wall times.  python tools/r5_genome_gpus.py [scale]   (scale 0.1: 25 records x 10^7 bp = 2.4e8 k-mers)"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import shapes
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
d = "/dev/shm/kc_r5gen"; os.makedirs(d, exist_ok=True)
recs = shapes.genome_records(25, int(1e8 * scale), 7)
with open(d + "/g.fa", "wb") as f:
    for i, r in enumerate(recs):
        f.write(b">chr%d\n" % i); f.write(shapes.ACGT[r].tobytes()); f.write(b"\n")
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""): h.update(blk)
    return h.hexdigest()
ref = None
for gp in (None, "2", "4", "8"):
    out = d + "/o.txt"
    if os.path.exists(out): os.unlink(out)
    time.sleep(3)
    t0 = time.time()
    p = subprocess.run([exe, "compute", "-k", "31"] + (["--gpus", gp] if gp else []) + ["-o", out, d + "/g.fa"], capture_output=True, text=True, env=dict(os.environ, KC_SHARD_TRACE="1" if gp else "0"))
    dt = time.time() - t0
    m = md5(out) if p.returncode == 0 else None
    if ref is None: ref = m
    steps = [l.split("step ")[1][:60] for l in p.stderr.splitlines() if "[shard 0/" in l and " step " in l]
    print("gpus %s: wall %.2f s rc %d identical %s %s" % (gp, dt, p.returncode, m == ref, " | ".join(steps)[:900]), flush=True)
    if p.returncode: print(p.stderr[-500:])
import shutil; shutil.rmtree(d)
