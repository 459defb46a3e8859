#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/ubench/pair_round 2>&1 | tee gpurun_out/r03_pair_round_ubench.txt
timeout -k 10 600 python -m pytest tests/test_gpu_nbody.py -m gpu -q -x -p no:cacheprovider -k "direct" -s > gpurun_out/r3g_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "direct n=|rel err|passed|failed|Error" gpurun_out/r3g_pytest.log | tail -n 12
if [ $rc -ne 0 ]; then tail -n 30 gpurun_out/r3g_pytest.log; exit 1; fi
for v in 1 0; do
  NBMI_DIRECT_SCALAR=$v timeout -k 10 300 python bench.py --workload cluster_1m_direct --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('scalar=$v', d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
