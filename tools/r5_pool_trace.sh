#!/bin/bash
mkdir -p gpurun_out/r5
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
KC_POOL_TRACE=1 KC_CLI_TRACE=1 kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_r5/out.txt /dev/shm/kc_r5/in.fa 2>&1 | grep "pool\]\|\[cli\]" > gpurun_out/r5/pool_trace.txt
rm -rf /dev/shm/kc_r5
cat gpurun_out/r5/pool_trace.txt
