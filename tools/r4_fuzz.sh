#!/bin/bash
mkdir -p gpurun_out/step
timeout 460 python3 tools/fuzz_parity.py 380 4104 > gpurun_out/step/fuzz_long.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/step/fuzz_long.log; tail -3 gpurun_out/step/fuzz_long.log
timeout 200 python3 tools/fuzz_cli_gpus.py 120 78 > gpurun_out/step/fuzz_gpus.log 2>&1
echo "fuzz gpus rc=$?" >> gpurun_out/step/fuzz_gpus.log; tail -3 gpurun_out/step/fuzz_gpus.log
