#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py tests/test_gpu_sharded_record.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3k_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 4 gpurun_out/r3k_pytest.log
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/r3k_pytest.log; exit 1; fi
run() {
  timeout -k 10 300 python bench.py --workload $1 --skip-10m --no-cpu-baseline --steps $2 --warmup 5 ${3:-} 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$1 ${3:-}', 'balance', os.environ.get('NBMI_XCD_BALANCE'), 'fp', os.environ.get('NBMI_FORCE_PREC'), 'ms', round(d['ms_per_step'],4), 'walk', round(d['phase_ms']['walk_ms'],4))"
}
for b in 1 0; do NBMI_XCD_BALANCE=$b run galaxy_1m_bh 20; done
for b in 1 0; do NBMI_XCD_BALANCE=$b run collision_10m_bh 8; done
for b in 1 0; do NBMI_FORCE_PREC=1 NBMI_XCD_BALANCE=$b run collision_10m_bh 8; done
for b in 1 0; do NBMI_XCD_BALANCE=$b run galaxy_1m_bh 10 "--bodies-per-gpu 4000000"; done
for b in 1 0; do NBMI_FORCE_PREC=1 NBMI_XCD_BALANCE=$b run galaxy_1m_bh 20; done
