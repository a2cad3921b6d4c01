#!/bin/bash
# round 5: step times of the sharded flow, ranks = threads sharing the one GPU (CLI --gpus N), cfg[1]-sized input
mkdir -p gpurun_out/r5
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5s"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 100000, 1030, fasta=d + "/in.fa")
PY
for G in 8 4; do
  for i in 1 2; do
    KC_SHARD_TRACE=1 kmercamel_amd/bin/kmercamel compute -k 31 --gpus $G -o /dev/shm/kc_r5s/out$G.txt /dev/shm/kc_r5s/in.fa 2>&1 | grep "step\|entries=" | grep "shard 0" > gpurun_out/r5/shard_trace_G$G.txt
  done
  echo "== G=$G"; cat gpurun_out/r5/shard_trace_G$G.txt
done
kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_r5s/out1.txt /dev/shm/kc_r5s/in.fa 2>/dev/null
cmp /dev/shm/kc_r5s/out1.txt /dev/shm/kc_r5s/out8.txt && cmp /dev/shm/kc_r5s/out1.txt /dev/shm/kc_r5s/out4.txt && echo "outputs identical"
rm -rf /dev/shm/kc_r5s
