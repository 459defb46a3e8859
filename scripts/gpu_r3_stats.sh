#!/bin/bash
# rocprofv3 kernel stats of the 1 M and 10 M bench runs (TAG names the output directory under gpurun_out/)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r03stats}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-galaxy_1m_bh collision_10m_bh}; do
  rm -rf $O/stats_$w
  extra=""; [ "$w" = galaxy_1m_bh ] && extra="--skip-10m"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline $extra > $O/stats_$w.log 2>&1
  rc=$?; echo "stats $w rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep '^{' $O/stats_$w.log > $O/${w}_bench_under_rocprof.json
  f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv
  python3 - "$O/${w}_kernel_stats.csv" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-60s calls=%5s avg_us=%9.1f"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf $O/stats_$w
done
