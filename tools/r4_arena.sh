#!/bin/bash
# arena size check: parity subset, CLI pool peak, default bench
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4_arena_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_arena_tests.log
tail -3 gpurun_out/r4_arena_tests.log
timeout 900 python3 bench.py > gpurun_out/r4_arena_bench.json 2> gpurun_out/r4_arena_bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r4_arena_bench.json
grep -i "pool" gpurun_out/r4_arena_bench.err | tail -5
