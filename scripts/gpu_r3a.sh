#!/bin/bash
# round 3, call A: precision diagnostics at 1 M bodies x 100 steps against the cached oracle, then the GPU suite
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/r3a_nproc.txt
MODES=${MODES:-0,1,2,3,4,4:16,5,6,7,8,9} timeout -k 10 900 python scripts/gpu_prec_diag.py > gpurun_out/prec_diag.jsonl 2> gpurun_out/prec_diag.err
rc=$?
echo "prec_diag rc=$rc"; tail -n 3 gpurun_out/prec_diag.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
grep '"step": 100' gpurun_out/prec_diag.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3a_pytest.log 2>&1
echo "pytest rc=$?"; tail -n 5 gpurun_out/r3a_pytest.log
