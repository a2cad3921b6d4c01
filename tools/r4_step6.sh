#!/bin/bash
mkdir -p gpurun_out/step
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -m gpu -x -q -k "segment_parallel or fragment_contraction or spneumoniae or shapes or full_size_properties_cfg1" > gpurun_out/step/t1.log 2>&1
echo "t1 rc=$?" >> gpurun_out/step/t1.log; tail -4 gpurun_out/step/t1.log
timeout 600 python3 tools/shapes.py genome --scale 1.0 --passes 3 --check > gpurun_out/step/genome.json 2> gpurun_out/step/genome.err
echo "genome rc=$?"; cut -c1-420 gpurun_out/step/genome.json
