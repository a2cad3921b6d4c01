#!/bin/bash
mkdir -p gpurun_out/step
timeout 560 python3 tools/fuzz_parity.py 480 20261004 > gpurun_out/step/fuzz_long.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/step/fuzz_long.log; tail -3 gpurun_out/step/fuzz_long.log
timeout 400 python3 tools/fuzz_cli_gpus.py 300 77 > gpurun_out/step/fuzz_gpus.log 2>&1
echo "fuzz gpus rc=$?" >> gpurun_out/step/fuzz_gpus.log; tail -3 gpurun_out/step/fuzz_gpus.log
