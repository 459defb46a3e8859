#!/bin/bash
# Sweep walk tuning knobs with short bench runs (one process per setting).
# SETTINGS: ';'-separated env assignments, e.g. "NBMI_XCD_CHUNK=0;NBMI_XCD_CHUNK=64;NBMI_WALK_BLOCK=128"
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep.log
IFS=';' read -ra SETS <<< "${SETTINGS:-NBMI_XCD_CHUNK=0}"
for setting in "${SETS[@]}"; do
  echo "=== $setting" | tee -a gpurun_out/sweep.log
  env $setting timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g ms/step %.3f phases %s'%(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['phase_ms'].items()}))
" | tee -a gpurun_out/sweep.log
  rc=${PIPESTATUS[0]}
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
done
