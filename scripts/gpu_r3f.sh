#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_record.py tests/test_gpu_boids.py -m gpu -q -x -p no:cacheprovider -s > gpurun_out/r3f_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "owner mode|rank [0-9]|passed|failed|Error" gpurun_out/r3f_pytest.log | tail -n 20
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/r3f_pytest.log; exit 1; fi
timeout -k 10 300 python scripts/gpu_owner_w1.py 2> gpurun_out/r3f_w1.err | tee gpurun_out/r3f_owner_w1.json
NBMI_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-10m 2> gpurun_out/r3f_dist1.err | cut -c1-400
