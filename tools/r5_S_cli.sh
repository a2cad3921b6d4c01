#!/bin/bash
# round 5: `kmercamel compute -S` (simplitigs_from_fasta) on the cfg[2] records taken as simplitigs (SURVEY 8 row B5 at the benchmark's size)
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5s2"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
for i in 1 2; do
  sleep 4; rm -f /dev/shm/kc_r5s2/out.txt
  python3 - <<'PY'
import subprocess, time, os
t0 = time.time()
p = subprocess.run(["kmercamel_amd/bin/kmercamel", "compute", "-k", "31", "-S", "-o", "/dev/shm/kc_r5s2/out.txt", "/dev/shm/kc_r5s2/in.fa"], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
print("compute -S: wall %.3f s rc %d" % (time.time() - t0, p.returncode), [l for l in p.stderr.splitlines() if "[cli]" in l and "epoch" not in l])
PY
done
ls -la /dev/shm/kc_r5s2/out.txt
rm -rf /dev/shm/kc_r5s2
