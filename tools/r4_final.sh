#!/bin/bash
# end of round: the whole GPU suite, then the evidence collection
mkdir -p gpurun_out/step
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/step/full.log 2>&1
echo "full rc=$?" >> gpurun_out/step/full.log; tail -4 gpurun_out/step/full.log
bash tools/collect_all_r04.sh
