"""round 5: the CLI end to end on a read set with coverage (150 bp reads, a header line each -- header-dense input for the device tokeniser),
with and without -z 2.  python tools/r5_reads_cli.py [genome_Mbp] [coverage]"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gm = float(sys.argv[1]) if len(sys.argv) > 1 else 100
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(7)
G, rl, er = int(gm * 1e6), 150, 0.01
n_reads = int(G * cov / rl)
genome = rng.integers(0, 4, size=G, dtype=np.uint8)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
d = "/dev/shm/kc_r5reads"; os.makedirs(d, exist_ok=True)
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
        a[:, :len(hdr)] = hdr
        a[:, len(hdr):len(hdr) + rl] = acgt[codes]
        a[:, -1] = 10
        f.write(a.tobytes())
print("file %.2f GB, %d reads" % (os.path.getsize(d + "/reads.fa") / 1e9, n_reads), flush=True)
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
for extra in ([], ["-z", "2"], ["-z", "2"], []):
    out = d + "/o.txt"
    if os.path.exists(out): os.unlink(out)
    time.sleep(4)
    t0 = time.time()
    p = subprocess.run([exe, "compute", "-k", "31"] + extra + ["-o", out, d + "/reads.fa"], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
    dt = time.time() - t0
    tr = [l[6:] for l in p.stderr.splitlines() if l.startswith("[cli]") and "epoch" not in l and "pool:" not in l and "pages" not in l]
    log = [l for l in p.stderr.splitlines() if "Finished collecting" in l or "simplitigs)" in l or "(l=" in l]
    print("compute %s: wall %.3f s rc %d | %s | %s" % (" ".join(extra), dt, p.returncode, " ; ".join(tr), " ; ".join(x.split("] ")[-1] for x in log)), flush=True)
import shutil; shutil.rmtree(d)
