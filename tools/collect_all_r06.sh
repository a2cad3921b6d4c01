#!/bin/bash
# Round-6 evidence in one gpurun call: kernel stats + HBM counter passes (FETCH_SIZE; WRITE_SIZE + L2 hits, separate passes) for cfg2
# (2.5e9 31-mers), cfg1 (1e8) and the two wide-key shapes of configs[4] (k = 63, k = 127 at 1e8 k-mers), summarised into profiles/ on the
# box and copied to gpurun_out/profiles_r06/ (gpurun merges only gpurun_out/ back); request counters of the probe kernel; one rank of the
# sharded flow at the size it has at 8 ranks; the two real-data shapes at scale; the default bench line (with the CLI and the CPU baseline).
set -u
R=$(pwd)
mkdir -p gpurun_out/profiles_r06
run() {   # tag, bench args...
    local tag=$1; shift
    timeout 700 bash tools/collect_profile.sh gpurun_out/prof_$tag --steps 3 --warmup 1 --no-cpu-baseline --no-cli "$@" > gpurun_out/prof_$tag.log 2>&1
    cd $R
    python3 tools/summarize_profile.py gpurun_out/prof_$tag $tag >> gpurun_out/prof_$tag.log 2>&1
    cp profiles/${tag}_* gpurun_out/profiles_r06/ 2>/dev/null
    rm -rf gpurun_out/prof_$tag
}
run r06_cfg2_2p5G
run r06_cfg1 --records 100000
run r06_k63 --records 100000 -k 63
run r06_k127 --records 100000 -k 127
cd $R
KC_PMC_ARGS="--no-cpu-baseline --no-cli --steps 1 --warmup 0" timeout 600 bash tools/kernel_pmc.sh kc_probe_block "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" > gpurun_out/profiles_r06/r06_cfg2_probe_kernel_pmc.txt 2>&1
cd $R
timeout 300 bash tools/shard_one_stats.sh 312500 > gpurun_out/profiles_r06/r06_shard_one_312k.log 2>&1
cp gpurun_out/shard_one/*/*_kernel_stats.csv gpurun_out/profiles_r06/r06_shard_one_312k_kernel_stats.csv 2>/dev/null
cd $R
timeout 600 python3 tools/shapes.py genome --check --passes 2 > gpurun_out/profiles_r06/r06_shape_genome_2p5G.json 2> gpurun_out/profiles_r06/r06_shape_genome.err
timeout 600 python3 tools/shapes.py coverage --check --passes 2 > gpurun_out/profiles_r06/r06_shape_coverage_1G.json 2> gpurun_out/profiles_r06/r06_shape_coverage.err
for kk in 63 127; do
  timeout 300 python3 bench.py --records 100000 -k $kk --steps 10 --warmup 2 --no-cpu-baseline --no-cli > gpurun_out/profiles_r06/r06_k${kk}_bench_line_unprofiled.json 2>/dev/null
done
timeout 300 python3 bench.py --records 100000 --steps 20 --warmup 2 --no-cpu-baseline --no-cli > gpurun_out/profiles_r06/r06_cfg1_bench_line_unprofiled.json 2>/dev/null
timeout 900 python3 bench.py --check > gpurun_out/profiles_r06/r06_default_bench.json 2> gpurun_out/profiles_r06/r06_default_bench.err
tail -c 1500 gpurun_out/profiles_r06/r06_default_bench.json
