set -x
nproc; free -g; df -h /dev/shm /tmp; rocm-smi --showmeminfo vram | head; ls /opt/rocm/lib | grep -i rccl | head
python - <<'PY'
import ctypes, os
lib = ctypes.CDLL("/opt/rocm/lib/librccl.so")
print("rccl loaded", lib)
PY
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --records 2500000 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/probe_bench_2p5G.json 2> gpurun_out/probe_bench_2p5G.err; echo rc=$?
tail -3 gpurun_out/probe_bench_2p5G.err; cat gpurun_out/probe_bench_2p5G.json
