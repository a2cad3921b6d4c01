#!/bin/bash
# round 5: `kmercamel maskopt` on the 2.55 GB masked superstring of cfg[2] (SURVEY 8f row 1 at the benchmark's size)
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5m"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
B=kmercamel_amd/bin/kmercamel; D=/dev/shm/kc_r5m
$B compute -k 31 -o $D/ms.txt $D/in.fa 2>/dev/null
rm -f $D/in.fa
for alg in max-one min-one; do
  for i in 1 2; do
    rm -f $D/opt.txt; sleep 4
    python3 - "$alg" <<'PY'
import subprocess, sys, time
alg = sys.argv[1]
t0 = time.time()
p = subprocess.run(["kmercamel_amd/bin/kmercamel", "maskopt", "-t", alg, "-k", "31", "-o", "/dev/shm/kc_r5m/opt.txt", "/dev/shm/kc_r5m/ms.txt"], capture_output=True, text=True, env=dict(__import__("os").environ, KC_CLI_TRACE="1"))
print("maskopt -t %s: wall %.3f s rc %d" % (alg, time.time() - t0, p.returncode), [l for l in p.stderr.splitlines() if "[cli]" in l or "rror" in l][:8])
PY
  done
  python3 - <<'PY'
import numpy as np
a = np.fromfile("/dev/shm/kc_r5m/opt.txt", dtype=np.uint8); b = np.fromfile("/dev/shm/kc_r5m/ms.txt", dtype=np.uint8)
na, nb = np.flatnonzero(a == 10), np.flatnonzero(b == 10)
x, y = a[na[0] + 1:na[1]], b[nb[0] + 1:nb[1]]
print("   header:", bytes(a[:na[0]]).decode()[:100], "| same characters:", bool(((x | 32) == (y | 32)).all()), "ones", int(np.count_nonzero(x < 97)), "(input", int(np.count_nonzero(y < 97)), ")")
PY
done
rm -rf $D
