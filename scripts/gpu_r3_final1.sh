#!/bin/bash
# round-3 evidence, part 1: GPU suite, rocprofv3 kernel stats of the three bench workloads, the default bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -rA > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 3 $O/pytest_gpu.log
cd /tmp
for w in galaxy_1m_bh collision_10m_bh boids_2m; do
  rm -rf $O/stats_$w
  extra=""; [ "$w" = galaxy_1m_bh ] && extra="--skip-10m"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline $extra > $O/stats_$w.log 2>&1
  rc=$?; echo "stats $w rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep '^{' $O/stats_$w.log > $O/r03_${w}_bench_under_rocprof.json
  f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r03_${w}_kernel_stats.csv
  rm -rf $O/stats_$w
done
cd $R
timeout -k 10 900 python bench.py > $O/r03_default_bench.json 2> $O/default_bench.err
echo "bench rc=$?"
python3 - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03/r03_default_bench.json'))
print(d['value'], d['ms_per_step'], d['phase_ms'], d['roofline']['frac'], d['roofline']['kernel_ms'])
n=d['north_star_10m']; print(n['value'], n['ms_per_step'], n['phase_ms'], n['roofline']['frac'])
print(d['cpu_baseline']); print(n['cpu_baseline']); print(d['frame_pcie'])
PY
