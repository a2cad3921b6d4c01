#!/bin/bash
mkdir -p gpurun_out/diag
export KC_SIMP_TRACE=1 KC_SET_TRACE=1
timeout 900 bash tools/shape_stats.sh genome 1.0 > gpurun_out/diag/genome_stats.txt 2>&1
cp gpurun_out/shape_stats/run.log gpurun_out/diag/genome_run.log
rm -rf gpurun_out/shape_stats
tail -30 gpurun_out/diag/genome_stats.txt
