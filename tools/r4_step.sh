#!/bin/bash
# parity + shapes after a change of phase 1
mkdir -p gpurun_out/step
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -m gpu -x -q > gpurun_out/step/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/step/tests.log
tail -4 gpurun_out/step/tests.log
KC_SIMP_TRACE=1 timeout 600 python3 tools/shapes.py genome --scale 1.0 --passes 2 --check > gpurun_out/step/genome.json 2> gpurun_out/step/genome.err
echo "genome rc=$?"; cut -c1-600 gpurun_out/step/genome.json; grep "\[simp\]" gpurun_out/step/genome.err | tail -8
KC_SIMP_TRACE=1 timeout 600 python3 tools/shapes.py coverage --scale 1.0 --passes 2 --check > gpurun_out/step/coverage.json 2> gpurun_out/step/coverage.err
echo "coverage rc=$?"; cut -c1-600 gpurun_out/step/coverage.json; grep "\[simp\]" gpurun_out/step/coverage.err | tail -8
