#!/bin/bash
# round 5: I/O options on the GPU box + today's CLI phase trace at cfg[2]
mkdir -p gpurun_out/r5
tools/io_probe.bin 2.6 > gpurun_out/r5/io_probe.txt 2>&1
python3 - > gpurun_out/r5/cli_trace.txt 2>&1 <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5"; os.makedirs(d, exist_ok=True)
fa = d + "/in.fa"
bench.generate(2, 0, 2500000, 1030, fasta=fa)
exe = "kmercamel_amd/bin/kmercamel"
for i in range(3):
    time.sleep(6)
    t0 = time.perf_counter()
    p = subprocess.run([exe, "compute", "-k", "31", "-o", d + "/out.txt", fa], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
    print("wall %.3f s" % (time.perf_counter() - t0))
    print("\n".join(l for l in p.stderr.splitlines() if "[cli]" in l), flush=True)
import shutil; shutil.rmtree(d)
PY
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_limit or alternative_device_paths or spill" 2>&1 | tail -5 > gpurun_out/r5/pytest_subset.txt
python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "fails_alone or rccl or world1" 2>&1 | tail -5 >> gpurun_out/r5/pytest_subset.txt
cat gpurun_out/r5/io_probe.txt gpurun_out/r5/cli_trace.txt gpurun_out/r5/pytest_subset.txt
