#!/bin/bash
# default bench line (what the driver runs) + the other configurations' lines        ROUND=r04 bash scripts/gpu_round_bench.sh
set -u
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-rXX}
O=$R/gpurun_out/profiles_$ROUND
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/${ROUND}_default_bench.json 2> gpurun_out/bench_err.txt || { tail -n 5 gpurun_out/bench_err.txt; exit 1; }
cut -c1-600 $O/${ROUND}_default_bench.json
: > $O/${ROUND}_other_configs_bench.jsonl
run() { timeout -k 10 600 python bench.py "$@" 2>> gpurun_out/bench_err.txt | tee -a $O/${ROUND}_other_configs_bench.jsonl | cut -c1-330; }
run --workload cluster_1m_direct --steps 3 --warmup 1
run --workload boids_2m
run --workload galaxy_10k_bh --steps 100 --warmup 10 --no-cpu-baseline
run --workload galaxy_10k_bh --bodies-per-gpu 100000 --steps 100 --warmup 10 --no-cpu-baseline
run --no-cpu-baseline --skip-10m --force-precision f32
run --no-cpu-baseline --skip-10m --force-precision f64
run --no-cpu-baseline --skip-10m --dt 0.01
run --no-cpu-baseline --skip-10m --theta 0.8
run --no-cpu-baseline --skip-10m --theta 1.3
run --workload collision_10m_bh --steps 10 --warmup 2 --no-cpu-baseline --force-precision f32
