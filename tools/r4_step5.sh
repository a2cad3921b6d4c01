#!/bin/bash
mkdir -p gpurun_out/step
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/step/t1.log 2>&1
echo "t1 rc=$?" >> gpurun_out/step/t1.log; tail -4 gpurun_out/step/t1.log
timeout 300 python3 tools/fuzz_parity.py 120 99 > gpurun_out/step/fuzz.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/step/fuzz.log; tail -2 gpurun_out/step/fuzz.log
KC_SET_TRACE=1 timeout 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-cli -k 127 --records 100000 2> gpurun_out/step/k127.err | tail -1 | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('k127',d['ms_per_step'],d['phases_ms'])"
grep "\[set\]" gpurun_out/step/k127.err | tail -1
