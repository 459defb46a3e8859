#!/bin/bash
# small systems: small emit tile + graded split walk; parity suite first, then step times at 10 k ... 262 k bodies
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py -m gpu -q -x -p no:cacheprovider -k "not 1m and not 10m" 2>&1 | tail -n 3
NBMI_SPLIT_GRADED=1 timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py -m gpu -q -x -p no:cacheprovider -k "not 1m and not 10m" 2>&1 | tail -n 3
for n in 10000 30000 100000 262144; do
for cfg in "prev 0" "new 0" "new 1"; do
  set -- $cfg
  if [ "$1" = prev ]; then export NBMI_LIB=$GRAFT_REPO_ROOT/3d-spatial-sim-for-boid-and-nbody_amd/libnbmi_prev.so; else unset NBMI_LIB; fi
  NBMI_SPLIT_GRADED=$2 timeout -k 10 300 python bench.py --workload galaxy_10k_bh --bodies-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r4m.json 2> gpurun_out/r4m_err.txt || { tail -n 5 gpurun_out/r4m_err.txt; exit 1; }
  python3 - $n $1 $2 <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r4m.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], 'graded', sys.argv[3], round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['phase_ms'].items()})
PY
done
done
