#!/bin/bash
# round 5: `kmercamel lowerbound` at cfg[2] (SURVEY 8f row 3 at the benchmark's size)
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5l"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
for v in "" "KC_CLI_NOSTREAM=1"; do
  for i in 1 2; do
    sleep 4
    python3 - "$v" <<'PY'
import subprocess, sys, time, os
env = dict(os.environ)
if sys.argv[1]: env[sys.argv[1].split("=")[0]] = "1"
t0 = time.time()
p = subprocess.run(["kmercamel_amd/bin/kmercamel", "lowerbound", "-k", "31", "/dev/shm/kc_r5l/in.fa"], capture_output=True, text=True, env=env)
print("lowerbound %s: wall %.3f s rc %d -> %s" % (sys.argv[1] or "(streamed)", time.time() - t0, p.returncode, p.stdout.strip()[:40]))
PY
  done
done
rm -rf /dev/shm/kc_r5l
