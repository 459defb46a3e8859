#!/bin/bash
# The round's evidence in one go: rocprofv3 kernel stats + PMC passes of the three headline workloads, copied into
# profiles/<ROUND>_* inside gpurun_out/profiles_<ROUND>/ (merge them into profiles/ afterwards), then the default bench line
# and the bench lines of the other configurations.      ROUND=r04 bash scripts/gpu_round_profiles.sh
set -u
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-rXX}
O=$R/gpurun_out/profiles_$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in galaxy_1m_bh collision_10m_bh boids_2m; do
  X=""; [ "$w" = galaxy_1m_bh ] && X="--skip-10m"   # (the default line's nested 10 M object would mix into the 1 M averages)
  rm -rf $R/gpurun_out/stats_$w
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --steady-steps 0 $X > $R/gpurun_out/stats_$w.log 2>&1
  rc=$?; echo "stats $w rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
  grep '^{' $R/gpurun_out/stats_$w.log > $O/${ROUND}_${w}_bench_under_rocprof.json
  cp $R/gpurun_out/stats_$w/*/*kernel_stats.csv $O/${ROUND}_${w}_kernel_stats.csv
  # per-dispatch durations of the build kernels (is a kernel's maximum its first dispatch?)
  python3 - $R/gpurun_out/stats_$w $w >> $O/${ROUND}_dispatch_spread.txt <<'PY'
import csv, glob, sys, collections
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    for k in ('k_gather_scan','k_emit_tile','k_keys','k_tiefix','k_xcd_bounds','k_walk<true','k_reorder','k_flock<true'):
        if k in n: d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    print(sys.argv[2], k, 'dispatches', len(v), 'first three', [round(x,1) for x in v[:3]], 'min', round(min(v),1), 'median', round(sorted(v)[len(v)//2],1), 'max', round(max(v),1), 'max at dispatch', v.index(max(v)))
    if k in ('k_walk<true', 'k_flock<true'):
        # bench.py --steps 10 --warmup 2: dispatches 0-1 warm-up, 2-11 the timed steps, 12-21 the same steps again under HIP events
        # (the figure the line prints as roofline.kernel_ms); what follows is the PCIe-inclusive frame leg, later in the run
        print(sys.argv[2], k, 'mean of the HIP-event pass (dispatches 12-21)', round(sum(v[12:22]) / 10, 1), 'us; all dispatches in order:', [round(x) for x in v])
PY
  cd $R
  TAG=${ROUND}_$w BENCH_ARGS="--workload $w --steady-steps 0 $X" bash scripts/gpu_pmc.sh > gpurun_out/pmc_${ROUND}_$w.log 2>&1 || { tail -n 5 gpurun_out/pmc_${ROUND}_$w.log; exit 1; }
  cp gpurun_out/pmc_${ROUND}_$w/summary.json $O/${ROUND}_${w}_pmc_summary.json
  rm -rf gpurun_out/pmc_${ROUND}_$w/pass* gpurun_out/stats_$w
  cd /tmp
done
cd $R
cat $O/${ROUND}_dispatch_spread.txt
