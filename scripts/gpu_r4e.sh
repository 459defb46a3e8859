#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python scripts/gpu_prec_cases.py 2>gpurun_out/r4e_err.txt | tee gpurun_out/r4e_prec_cases.jsonl | cut -c1-420
tail -n 3 gpurun_out/r4e_err.txt
