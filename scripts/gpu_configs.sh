#!/bin/bash
# Bench lines for the other BASELINE configs (3: direct 1M, 4: collision 10M on one GPU, 5: boids 2M).
set -u
mkdir -p gpurun_out
for w in ${WORKLOADS:-boids_2m cluster_1m_direct collision_10m_bh}; do
  steps=10; [ "$w" = cluster_1m_direct ] && steps=3
  echo "=== $w" | tee -a gpurun_out/configs.log
  timeout -k 10 ${T:-500} python bench.py --workload $w --steps $steps --warmup 1 ${EXTRA:-} 2> gpurun_out/cfg_$w.err | tee -a gpurun_out/configs.log | cut -c1-1800
  rc=${PIPESTATUS[0]}
  echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
done
