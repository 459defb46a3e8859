#!/bin/bash
# round 3, call C: float64 force loop - modes (error, time), then the new / tightened GPU tests
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
N10M=1 timeout -k 10 900 python scripts/gpu_prec_modes.py > gpurun_out/prec_modes.jsonl 2> gpurun_out/prec_modes.err
rc=$?
echo "prec_modes rc=$rc"; tail -n 5 gpurun_out/prec_modes.err; cat gpurun_out/prec_modes.jsonl
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider -s -k "not sharded" > gpurun_out/r3c_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "1 M x|force precision|20 k bodies|passed|failed|Error|error" gpurun_out/r3c_pytest.log | tail -n 20
