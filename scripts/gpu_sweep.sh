#!/bin/bash
# Sweep walk tuning knobs with short bench runs (one process per setting).
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for cur in ${CURSORS:-1 2 4}; do
  for remap in ${REMAPS:-1 0}; do
   for wb in ${BLOCKS:-256}; do
    echo "=== cursors=$cur remap=$remap block=$wb" | tee -a gpurun_out/sweep.log
    NBMI_WALK_BLOCK=$wb NBMI_WALK_CURSORS=$cur NBMI_XCD_REMAP=$remap timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g ms/step %.3f phases %s'%(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['phase_ms'].items()}))
" | tee -a gpurun_out/sweep.log
    rc=${PIPESTATUS[0]}
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
   done
  done
done
