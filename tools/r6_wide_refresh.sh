#!/bin/bash
# round 6: the wide-key evidence again after the register-spill fix (kernel stats + traffic + unprofiled bench lines of k = 63 / 127)
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r06
run() {
    local tag=$1; shift
    timeout 700 bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > gpurun_out/prof_$tag.log 2>&1
    cd $R
    python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
    cp profiles/${tag}_* gpurun_out/profiles_r06/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
run r06_k63 --records 100000 -k 63
run r06_k127 --records 100000 -k 127
cd $R
for kk in 63 127; do
  timeout 300 python3 bench.py --records 100000 -k $kk --steps 10 --warmup 2 --no-cpu-baseline --no-cli --check > gpurun_out/profiles_r06/r06_k${kk}_bench_line_unprofiled.json 2>/dev/null
done
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -q -m gpu -k "wide_keys or simplitigs_match_canonical_oracle or ranks_wide_keys or self_comm" 2>&1 | tail -3
