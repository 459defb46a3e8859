#!/bin/bash
set -u
mkdir -p gpurun_out
SOAK_BOIDS=0 timeout -k 10 600 python scripts/gpu_soak.py 2>gpurun_out/r4k_err.txt | tee gpurun_out/r4k_soak.txt
tail -n 5 gpurun_out/r4k_err.txt
