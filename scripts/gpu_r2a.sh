#!/bin/bash
# Round 2, first GPU session: the new parity tests, then the bench with and without the per-lane yardstick.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/summary.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/summary.log
  tail -n ${TAILN:-30} "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name - stopping"; exit 1; fi
  return $rc
}
: > gpurun_out/summary.log
nproc | tee -a gpurun_out/summary.log
run smoke 600 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
run pytest_new 900 python -m pytest tests/test_gpu_nbody.py -m gpu -q -rA -x -p no:cacheprovider -k "${PYTEST_K:-sticky or long_run or accept_sets or north_star or cluster_1m or update_against or capacity or galaxy_1m or every_walk}" 
run bench1 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-10m
NBMI_WALK_PAIR=0 run bench1_nopair 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-10m
NBMI_WALK_LANE=1 run bench1_lane 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-10m
exit 0
