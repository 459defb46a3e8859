#!/bin/bash
# round 4, call 1: parity suite with the speculative next-record fetch, then A/B against the build without it
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r4a_pytest.log 2>&1 || { tail -n 30 gpurun_out/r4a_pytest.log; exit 1; }
tail -n 3 gpurun_out/r4a_pytest.log
bash scripts/gpu_ab_lib.sh 3d-spatial-sim-for-boid-and-nbody_amd/libnbmi_nospec.so 2>&1 | tee gpurun_out/r4a_ab.txt
