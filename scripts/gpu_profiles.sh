#!/bin/bash
# Kernel-trace stats + PMC passes for the default bench (galaxy 1M) and the boids workload.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in galaxy_1m_bh boids_2m; do
  rm -rf $R/gpurun_out/stats_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/stats_$w.log 2>&1
  rc=$?; echo "stats $w rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep '^{' $R/gpurun_out/stats_$w.log > $R/gpurun_out/stats_$w.json
done
cd $R
TAG=final_galaxy_1m_bh bash scripts/gpu_pmc.sh > gpurun_out/pmc_final_galaxy.log 2>&1 || exit 1
TAG=final_boids_2m BENCH_ARGS="--workload boids_2m" bash scripts/gpu_pmc.sh > gpurun_out/pmc_final_boids.log 2>&1 || exit 1
tail -3 gpurun_out/pmc_final_galaxy.log | cut -c1-300
