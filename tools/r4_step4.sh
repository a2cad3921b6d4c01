#!/bin/bash
mkdir -p gpurun_out/step
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -m gpu -x -q > gpurun_out/step/t1.log 2>&1
echo "t1 rc=$?" >> gpurun_out/step/t1.log; tail -5 gpurun_out/step/t1.log
timeout 300 python3 tools/fuzz_parity.py 150 31337 > gpurun_out/step/fuzz.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/step/fuzz.log; tail -3 gpurun_out/step/fuzz.log
if grep -q "t1 rc=0" gpurun_out/step/t1.log; then
KC_OV_TRACE=1 timeout 600 python3 tools/shapes.py coverage --scale 1.0 --passes 2 --check > gpurun_out/step/coverage.json 2> gpurun_out/step/coverage.err
echo "coverage rc=$?"; cut -c1-420 gpurun_out/step/coverage.json; grep "sample of" gpurun_out/step/coverage.err | tail -2
timeout 600 bash tools/shape_stats.sh coverage 1.0 > gpurun_out/step/coverage_stats.txt 2>&1; rm -rf gpurun_out/shape_stats
head -24 gpurun_out/step/coverage_stats.txt
timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/step/bench.json 2>gpurun_out/step/bench.err; cut -c1-900 gpurun_out/step/bench.json
fi
