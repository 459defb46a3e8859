#!/bin/bash
# One gpurun call: smoke -> GPU parity tests -> short bench.  Stops after any step that was
# killed by its timeout (never start another GPU step after a hang).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/summary.log
  timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/summary.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name - stopping"; exit 1; fi
  return $rc
}
: > gpurun_out/summary.log
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 | tee -a gpurun_out/summary.log
nproc | tee -a gpurun_out/summary.log
run smoke 600 python -c "import __graft_entry__ as g; g.smoke()"
run pytest_gpu ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -rA -p no:cacheprovider ${PYTEST_ARGS:-}
run bench 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2
if [ -n "${PROFILE:-}" ]; then
  cd /tmp
  run_dir=$GRAFT_REPO_ROOT/gpurun_out/prof_$PROFILE
  rm -rf "$run_dir"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$run_dir" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
  echo "=== rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/gpurun_out/summary.log
  cd $GRAFT_REPO_ROOT
  find "$run_dir" -name "*stats*.csv" | head
fi
exit 0
