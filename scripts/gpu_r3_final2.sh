#!/bin/bash
# round-3 evidence, part 2: PMC passes (1 M default, 10 M, boids) and the bench lines of the other configs / modes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
TAG=r03_galaxy_1m_bh BENCH_ARGS="--skip-10m" bash scripts/gpu_pmc.sh > $O/pmc_galaxy.log 2>&1 || { echo pmc galaxy failed; tail -5 $O/pmc_galaxy.log; exit 1; }
cp gpurun_out/pmc_r03_galaxy_1m_bh/summary.json $O/r03_galaxy_1m_bh_pmc_summary.json
TAG=r03_collision_10m_bh BENCH_ARGS="--workload collision_10m_bh" bash scripts/gpu_pmc.sh > $O/pmc_collision.log 2>&1 || { echo pmc collision failed; exit 1; }
cp gpurun_out/pmc_r03_collision_10m_bh/summary.json $O/r03_collision_10m_bh_pmc_summary.json
TAG=r03_boids_2m BENCH_ARGS="--workload boids_2m" bash scripts/gpu_pmc.sh > $O/pmc_boids.log 2>&1 || { echo pmc boids failed; exit 1; }
cp gpurun_out/pmc_r03_boids_2m/summary.json $O/r03_boids_2m_pmc_summary.json
rm -rf gpurun_out/pmc_r03_*/pass*
: > $O/r03_other_configs_bench.jsonl
for args in "--workload cluster_1m_direct --steps 3 --warmup 1" "--workload boids_2m --steps 30 --warmup 3" "--workload galaxy_10k_bh --steps 50 --warmup 5" \
            "--skip-10m --dt 0.01" "--skip-10m --force-precision f32" "--skip-10m --force-precision f64" \
            "--workload collision_10m_bh --force-precision f32 --steps 8 --warmup 2" "--skip-10m --theta 0.8" "--skip-10m --theta 1.3"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | grep '^{' >> $O/r03_other_configs_bench.jsonl
  echo "bench $args rc=$?"
done
python3 - <<'PY'
import json,os
for l in open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03/r03_other_configs_bench.jsonl'):
    d=json.loads(l); print(d['config'].get('workload'), d['config'].get('force_precision'), d['config'].get('theta'), d['config'].get('dt'), round(d['ms_per_step'],4), '%.3e'%d['value'])
PY
tail -2 $O/pmc_galaxy.log | cut -c1-400
