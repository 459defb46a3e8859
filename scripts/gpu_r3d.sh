#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py tests/test_gpu_sort.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3d_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 6 gpurun_out/r3d_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
TAG=${TAG:-r03stats_b} bash scripts/gpu_r3_stats.sh
