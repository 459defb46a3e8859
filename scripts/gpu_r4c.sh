#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 scripts/ubench/direct_mfma 1.0 2>&1 | tee gpurun_out/r4c_direct_mfma.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_parity_long.py -m gpu -q -x -s -p no:cacheprovider -k "collision_10m" > gpurun_out/r4b_10m.log 2>&1 || { tail -n 40 gpurun_out/r4b_10m.log; exit 1; }
grep -E "collision 10 M|passed|failed|skipped" gpurun_out/r4b_10m.log
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_record.py -m gpu -q -x -s -p no:cacheprovider > gpurun_out/r4b_sharded.log 2>&1 || { tail -n 40 gpurun_out/r4b_sharded.log; exit 1; }
grep -E "owner mode, 1 M|single handle|passed|failed" gpurun_out/r4b_sharded.log
