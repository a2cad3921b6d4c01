#!/bin/bash
# round 5: what one collective costs on the RCCL transport (one rank, its own share through grouped ncclSend / ncclRecv to itself), cfg[1]-sized input,
# sampled ranking forced so that the slotted exchanges run (headers travel even when no item does)
mkdir -p gpurun_out/r5
python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5r"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 100000, 1030, fasta=d + "/in.fa")
PY
for sr in 1 0; do
KC_SHARD_TRACE=1 KC_RCCL_SELF_SENDRECV=$sr KC_RANK_SAMPLED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29871 tests/shard_worker.py /dev/shm/kc_r5r/in.fa 31 bi /dev/shm/kc_r5r/out.npy --backend nccl 2>&1 | grep "step\|sharded ok\|sampled" > gpurun_out/r5/rccl_latency_self$sr.txt
echo "== KC_RCCL_SELF_SENDRECV=$sr"; cat gpurun_out/r5/rccl_latency_self$sr.txt
done
rm -rf /dev/shm/kc_r5r
