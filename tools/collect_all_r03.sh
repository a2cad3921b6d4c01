#!/bin/bash
# Round-3 evidence in one gpurun call: stats + HBM counter passes for cfg2 (2.5e9 31-mers), cfg1 (1e8), k=63 and k=127, summarised into
# profiles/ on the box and copied to gpurun_out/profiles_r03/ (gpurun merges only gpurun_out/ back).  Then the default bench line.
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r03
run() {   # tag, bench args...
    local tag=$1; shift
    bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > gpurun_out/prof_$tag.log 2>&1
    cd $R
    python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
    cp profiles/${tag}_* gpurun_out/profiles_r03/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
run r03_cfg2_2p5G
run r03_cfg1 --records 100000
run r03_k63 -k 63 --records 100000
run r03_k127 -k 127 --records 100000
python3 bench.py > gpurun_out/profiles_r03/r03_default_bench.log 2>&1
tail -1 gpurun_out/profiles_r03/r03_default_bench.log
