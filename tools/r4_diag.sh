#!/bin/bash
# diagnostics: pool trace, per-level overlap trace and a kernel timeline of phase 2 at the default configuration
mkdir -p gpurun_out/diag
KC_POOL_TRACE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/diag/pool.json 2> gpurun_out/diag/pool.err
grep "\[pool\]" gpurun_out/diag/pool.err | head -60 > gpurun_out/diag/pool.txt
KC_OV_TRACE=1 timeout 600 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/diag/ov.json 2> gpurun_out/diag/ov.err
grep "^\[ov\]" gpurun_out/diag/ov.err | tail -80 > gpurun_out/diag/ov.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/ov_timeline.py /tmp/kt > gpurun_out/diag/ov_timeline.txt 2>&1
cat gpurun_out/diag/ov_timeline.txt | head -60
