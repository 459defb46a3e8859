#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
P=3d-spatial-sim-for-boid-and-nbody_amd
bash scripts/gpu_ab_boids.sh $P/libnbmi_b86.so $P/libnbmi_b94.so 2>&1 | tee gpurun_out/r3h_boids_ab.txt
for v in 2 4 0; do
  NBMI_DIRECT_SCALAR=$v timeout -k 10 300 python bench.py --workload cluster_1m_direct --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct scalar=$v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/r3h_direct.txt
timeout -k 10 300 python -m pytest tests/test_gpu_boids.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
NBMI_DIRECT_SCALAR=2 timeout -k 10 300 python -m pytest tests/test_gpu_nbody.py -m gpu -q -x -p no:cacheprovider -k direct 2>&1 | tail -3
