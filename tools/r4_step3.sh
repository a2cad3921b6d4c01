#!/bin/bash
mkdir -p gpurun_out/step
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "segment_parallel or fragment_contraction or spneumoniae" > gpurun_out/step/t1.log 2>&1
echo "t1 rc=$?" >> gpurun_out/step/t1.log; tail -5 gpurun_out/step/t1.log
timeout 600 python3 -m pytest tests/test_gpu_shapes.py -m gpu -x -q > gpurun_out/step/t2.log 2>&1
echo "t2 rc=$?" >> gpurun_out/step/t2.log; tail -3 gpurun_out/step/t2.log
timeout 300 python3 tools/fuzz_parity.py 150 1234 > gpurun_out/step/fuzz.log 2>&1
echo "fuzz rc=$?" >> gpurun_out/step/fuzz.log; tail -3 gpurun_out/step/fuzz.log
if grep -q "t1 rc=0" gpurun_out/step/t1.log && grep -q "t2 rc=0" gpurun_out/step/t2.log; then
KC_SIMP_TRACE=1 timeout 600 python3 tools/shapes.py genome --scale 1.0 --passes 2 --check > gpurun_out/step/genome.json 2> gpurun_out/step/genome.err
echo "genome rc=$?"; cut -c1-420 gpurun_out/step/genome.json
timeout 600 bash tools/shape_stats.sh genome 1.0 > gpurun_out/step/genome_stats.txt 2>&1; rm -rf gpurun_out/shape_stats
head -16 gpurun_out/step/genome_stats.txt
fi
