#!/bin/bash
# round 3, call B: the rebuilt tree build (fused gather+scan, tile emit) through the GPU suite, then precision modes
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3b_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 15 gpurun_out/r3b_pytest.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
WHERE=1 MODES=${MODES:-0,10,11,12,13,14,20:20,20:60,20:120} timeout -k 10 900 python scripts/gpu_prec_diag.py > gpurun_out/prec_diag_b.jsonl 2> gpurun_out/prec_diag_b.err
rc=$?
echo "prec_diag rc=$rc"; tail -n 3 gpurun_out/prec_diag_b.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT - stopping"; exit 1; fi
grep '"step": 100' gpurun_out/prec_diag_b.jsonl
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('gpurun_out/r3b_bench.json'));print(d['ms_per_step'],d['phase_ms'],d['north_star_10m']['ms_per_step'],d['north_star_10m']['phase_ms'])"
