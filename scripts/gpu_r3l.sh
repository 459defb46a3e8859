#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --workload $1 --skip-10m --no-cpu-baseline --steps $2 --warmup 5 ${3:-} 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$1 ${3:-}', 'balance', os.environ.get('NBMI_XCD_BALANCE'), 'fp', os.environ.get('NBMI_FORCE_PREC'), 'ms', round(d['ms_per_step'],4), 'walk', round(d['phase_ms']['walk_ms'],4))"
}
run galaxy_1m_bh 20
NBMI_XCD_BALANCE=2 run galaxy_1m_bh 20
NBMI_FORCE_PREC=1 run galaxy_1m_bh 20
run collision_10m_bh 8
NBMI_XCD_BALANCE=0 run collision_10m_bh 8
