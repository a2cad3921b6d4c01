#!/bin/bash
# Round-4 evidence in one gpurun call: stats + HBM counter passes for cfg2 (2.5e9 31-mers), cfg1 (1e8), k=63 and k=127, summarised into
# profiles/ on the box and copied to gpurun_out/profiles_r04/ (gpurun merges only gpurun_out/ back); one sharded rank; the real-data
# shapes at full scale; then the default bench line.
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r04
run() {   # tag, bench args...
    local tag=$1; shift
    timeout 600 bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > gpurun_out/prof_$tag.log 2>&1
    cd $R
    python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
    cp profiles/${tag}_* gpurun_out/profiles_r04/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
run r04_cfg2_2p5G
run r04_cfg1 --records 100000
run r04_k63 -k 63 --records 100000
run r04_k127 -k 127 --records 100000
timeout 300 bash tools/shard_one_stats.sh 1000000 > gpurun_out/profiles_r04/r04_shard_one.log 2>&1
cp gpurun_out/shard_one/*/*_kernel_stats.csv gpurun_out/profiles_r04/r04_shard_one_kernel_stats.csv 2>/dev/null
cd $R
timeout 600 python3 tools/shapes.py genome --check --passes 2 > gpurun_out/profiles_r04/r04_shape_genome_2p5G.json 2> gpurun_out/profiles_r04/r04_shape_genome.err
timeout 600 python3 tools/shapes.py coverage --check --passes 2 > gpurun_out/profiles_r04/r04_shape_coverage_1G.json 2> gpurun_out/profiles_r04/r04_shape_coverage.err
timeout 600 bash tools/shape_stats.sh genome 1.0 > gpurun_out/profiles_r04/r04_shape_genome_kernel_stats.txt 2>&1
timeout 600 bash tools/shape_stats.sh coverage 1.0 > gpurun_out/profiles_r04/r04_shape_coverage_kernel_stats.txt 2>&1
rm -rf gpurun_out/shape_stats
cd $R
timeout 300 python3 tools/time_coverage_reads.py > gpurun_out/profiles_r04/r04_coverage_10x.log 2>&1
KC_OV_TRACE=1 timeout 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli 2>&1 | grep "^\[ov\]" | tail -40 > gpurun_out/profiles_r04/r04_cfg2_overlap_levels.txt
timeout 900 python3 bench.py > gpurun_out/profiles_r04/r04_default_bench.log 2>&1
tail -1 gpurun_out/profiles_r04/r04_default_bench.log | cut -c1-600
