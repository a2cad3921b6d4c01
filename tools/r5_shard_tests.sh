#!/bin/bash
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "fragment_ranking_variants or exchange_volume or ranks_match_single_gpu or branching" -s 2>&1 | tail -15 | tee gpurun_out/r5/shard_tests.txt
tools/r5_shard_trace.sh
