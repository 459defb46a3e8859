#!/bin/bash
# Round-2 evidence, part 1: rocprofv3 kernel stats (1 M galaxy, 10 M collision, boids) and the PMC passes of the
# default bench and of the boids bench.  Everything lands under gpurun_out/r02/ for copying into profiles/.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in galaxy_1m_bh collision_10m_bh boids_2m; do
  rm -rf $O/stats_$w
  extra=""; [ "$w" = galaxy_1m_bh ] && extra="--skip-10m"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline $extra > $O/stats_$w.log 2>&1
  rc=$?; echo "stats $w rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep '^{' $O/stats_$w.log > $O/r02_${w}_bench_under_rocprof.json
  f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r02_${w}_kernel_stats.csv
done
cd $R
TAG=r02_galaxy_1m_bh BENCH_ARGS="--skip-10m" bash scripts/gpu_pmc.sh > $O/pmc_galaxy.log 2>&1 || exit 1
cp gpurun_out/pmc_r02_galaxy_1m_bh/summary.json $O/r02_galaxy_1m_bh_pmc_summary.json
TAG=r02_boids_2m BENCH_ARGS="--workload boids_2m" bash scripts/gpu_pmc.sh > $O/pmc_boids.log 2>&1 || exit 1
cp gpurun_out/pmc_r02_boids_2m/summary.json $O/r02_boids_2m_pmc_summary.json
tail -3 $O/pmc_galaxy.log | cut -c1-300
ls $O
