#!/bin/bash
# round 5: tmpfs page allocation options + kernel trace of the CLI at cfg[2]
mkdir -p gpurun_out/r5
tools/io_probe2.bin 2.55 > gpurun_out/r5/io_probe2.txt 2>&1
python3 - > gpurun_out/r5/gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, ".")
import bench
d = "/dev/shm/kc_r5"; os.makedirs(d, exist_ok=True)
bench.generate(2, 0, 2500000, 1030, fasta=d + "/in.fa")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/cli_prof -o cli -- $GRAFT_REPO_ROOT/kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_r5/out.txt /dev/shm/kc_r5/in.fa > $GRAFT_REPO_ROOT/gpurun_out/r5/cli_prof.log 2>&1
cd $GRAFT_REPO_ROOT
rm -rf /dev/shm/kc_r5
find gpurun_out/r5/cli_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {} > gpurun_out/r5/cli_kernel_stats_head.csv
find gpurun_out/r5/cli_prof -name "*kernel_trace.csv" -delete
cat gpurun_out/r5/io_probe2.txt; cat gpurun_out/r5/cli_kernel_stats_head.csv | cut -c1-160
