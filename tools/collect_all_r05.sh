#!/bin/bash
# Round-5 evidence in one gpurun call: kernel stats + HBM counter passes for cfg2 (2.5e9 31-mers) and cfg1 (1e8), summarised into profiles/
# on the box and copied to gpurun_out/profiles_r05/ (gpurun merges only gpurun_out/ back); the CLI under the kernel trace; one rank of the
# sharded flow at the size it has at 8 ranks; the sharded step trace; the default bench line.
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r05
run() {   # tag, bench args...
    local tag=$1; shift
    timeout 600 bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > gpurun_out/prof_$tag.log 2>&1
    cd $R
    python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
    cp profiles/${tag}_* gpurun_out/profiles_r05/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
run r05_cfg2_2p5G
run r05_cfg1 --records 100000
cd $R
timeout 300 bash tools/shard_one_stats.sh 312500 > gpurun_out/profiles_r05/r05_shard_one_312k.log 2>&1
cp gpurun_out/shard_one/*/*_kernel_stats.csv gpurun_out/profiles_r05/r05_shard_one_312k_kernel_stats.csv 2>/dev/null
cd $R
timeout 300 bash tools/r5_cliprof.sh > gpurun_out/profiles_r05/r05_cli_kernel_and_copy_stats.txt 2>&1
cd $R
timeout 300 bash tools/r5_shard_trace.sh > gpurun_out/profiles_r05/r05_shard_steps_thread_ranks.txt 2>&1
timeout 600 python3 tools/shapes.py genome --check --passes 2 > gpurun_out/profiles_r05/r05_shape_genome_2p5G.json 2> gpurun_out/profiles_r05/r05_shape_genome.err
timeout 600 python3 tools/shapes.py coverage --check --passes 2 > gpurun_out/profiles_r05/r05_shape_coverage_1G.json 2> gpurun_out/profiles_r05/r05_shape_coverage.err
timeout 900 python3 bench.py > gpurun_out/profiles_r05/r05_default_bench.json 2> gpurun_out/profiles_r05/r05_default_bench.err
tail -c 1500 gpurun_out/profiles_r05/r05_default_bench.json
