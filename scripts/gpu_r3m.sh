#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
MODES=${MODES1M:-auto:5e-5,auto:1e-4} timeout -k 10 600 python scripts/gpu_prec_modes.py 2>/dev/null | tee gpurun_out/prec_modes_q.jsonl | cut -c1-420
MODES=${MODES10M:-auto:5e-5,auto:1e-3,f64} timeout -k 10 900 python scripts/gpu_prec_10m.py 2>/dev/null | tee gpurun_out/prec_10m_q.jsonl | cut -c1-600
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py tests/test_gpu_sharded_record.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
