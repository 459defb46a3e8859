#!/bin/bash
# N-body GPU tests with the in-tree library, then A/B of it against another build ($1, path relative to the repo)
set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/ab
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py tests/test_gpu_sharded_record.py -x -q -m gpu > gpurun_out/ab/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/ab/pytest.log
[ $rc -ne 0 ] && exit $rc
bash scripts/gpu_ab_lib.sh "$1"
