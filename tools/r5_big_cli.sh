#!/bin/bash
# round 5: the CLI end to end on 4.4e9 k-mers (a 4.5 GB FASTA file: beyond 2^32 bytes for the tokeniser, the mapping and the copies)
mkdir -p gpurun_out/r5
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5b"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 4400000, 1030, fasta=d + "/in.fa")
PY
for i in 1 2; do
  sleep 5; rm -f /dev/shm/kc_r5b/out.txt
  s=$(date +%s.%N); KC_CLI_TRACE=1 kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_r5b/out.txt /dev/shm/kc_r5b/in.fa 2> /dev/shm/kc_r5b/err.txt; e=$(date +%s.%N)
  python3 -c "print(\"wall %.3f s\" % ($e - $s))"; grep "\[cli\]" /dev/shm/kc_r5b/err.txt | grep -v epoch | tr '\n' '|'; echo
done
python3 - <<'PY'
import numpy as np
o = np.fromfile("/dev/shm/kc_r5b/out.txt", dtype=np.uint8)
nl = np.flatnonzero(o == 10)
print("lines", len(nl), "header", bytes(o[:nl[0]]).decode()[:80], "chars", nl[1] - nl[0] - 1, "ones", int(np.count_nonzero((o[nl[0]+1:nl[1]] < 97))))
PY
rm -rf /dev/shm/kc_r5b
