#!/bin/bash
# round 6, last call: kernel stats + traffic of cfg[2] and the default bench line (CLI + CPU baseline + --check) on the final build
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r06
tag=r06_cfg2_2p5G
timeout 700 bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/prof_$tag.log 2>&1
cd $R
python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
cp profiles/${tag}_* gpurun_out/profiles_r06/ 2>/dev/null
rm -rf gpurun_out/prof_$tag
timeout 900 python3 bench.py --check > gpurun_out/profiles_r06/r06_default_bench.json 2> gpurun_out/profiles_r06/r06_default_bench.err
tail -c 1200 gpurun_out/profiles_r06/r06_default_bench.json
