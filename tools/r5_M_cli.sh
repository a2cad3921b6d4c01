#!/bin/bash
# round 5: `kmercamel compute -M` at cfg[2] (two 2.55 GB files out)
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5M"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
for i in 1 2 3; do
  sleep 4; rm -f /dev/shm/kc_r5M/out.txt /dev/shm/kc_r5M/mask.txt
  python3 - <<'PY'
import subprocess, time, os
t0 = time.time()
p = subprocess.run(["kmercamel_amd/bin/kmercamel", "compute", "-k", "31", "-o", "/dev/shm/kc_r5M/out.txt", "-M", "/dev/shm/kc_r5M/mask.txt", "/dev/shm/kc_r5M/in.fa"], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
print("compute -M: wall %.3f s rc %d" % (time.time() - t0, p.returncode), [l[6:] for l in p.stderr.splitlines() if "[cli]" in l and "epoch" not in l and "pool" not in l])
PY
done
ls -la /dev/shm/kc_r5M/
rm -rf /dev/shm/kc_r5M
