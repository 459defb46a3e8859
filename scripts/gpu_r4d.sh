#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 scripts/ubench/direct_mfma 1.0 2>&1 | grep -v "^probe\|relative acc" | tee gpurun_out/r4d_direct_mfma.txt || exit 1
echo "--- built with -mllvm -amdgpu-mfma-vgpr-form=1" | tee -a gpurun_out/r4d_direct_mfma.txt
timeout -k 10 300 scripts/ubench/direct_mfma_vgpr 1.0 2>&1 | grep -v "^probe\|relative acc" | tee -a gpurun_out/r4d_direct_mfma.txt || exit 1
timeout -k 10 1100 python -m pytest tests/test_gpu_parity_long.py -m gpu -q -s -p no:cacheprovider -k "held_out" > gpurun_out/r4d_heldout.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 2
grep -E "steps:|passed|failed|skipped|Error|assert" gpurun_out/r4d_heldout.log | cut -c1-400
