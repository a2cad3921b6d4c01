#!/bin/bash
# round 5: `kmercamel compute --gpus N` end to end at cfg[2], the rank threads sharing the box's one GPU (what it costs beyond the kernels)
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5g"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, int(os.environ.get("KC_R5_RECORDS", "2500000")), 1030, fasta=d + "/in.fa")
PY
for v in "2" "2 KC_CLI_NOSTREAM=1" "4"; do
  set -- $v
  sleep 5; rm -f /dev/shm/kc_r5g/out.txt
  python3 - "$1" "${2:-}" <<'PY'
import subprocess, sys, time, os
env = dict(os.environ, KC_CLI_TRACE="1")
if sys.argv[2]: env[sys.argv[2].split("=")[0]] = "1"
t0 = time.time()
p = subprocess.run(["kmercamel_amd/bin/kmercamel", "compute", "-k", "31", "--gpus", sys.argv[1], "-o", "/dev/shm/kc_r5g/out.txt", "/dev/shm/kc_r5g/in.fa"], capture_output=True, text=True, env=env)
print("compute --gpus %s %s: wall %.3f s rc %d" % (sys.argv[1], sys.argv[2], time.time() - t0, p.returncode), [l[6:] for l in p.stderr.splitlines() if l.startswith("[cli]") and "epoch" not in l and "pool" not in l][:12], [l for l in p.stderr.splitlines() if not l.startswith("[")][-3:])
PY
  md5sum /dev/shm/kc_r5g/out.txt | cut -c1-32
done
rm -rf /dev/shm/kc_r5g
