#!/bin/bash
# boids in the state flocks reach: bench line with the steady_state object, then PMC passes of the last dispatches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python bench.py --workload boids_2m --no-cpu-baseline > gpurun_out/r4g_boids_bench.json 2> gpurun_out/r4g_err.txt || { tail -n 20 gpurun_out/r4g_err.txt; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4g_boids_bench.json').read().strip().splitlines()[-1])
print('t=0   ', round(d['ms_per_step'],4), d['phase_ms'], d['config'].get('candidates_per_boid'), d['roofline']['occupied_cells'])
s=d['steady_state']; print('steady', round(s['ms_per_step'],4), s['phase_ms'], s['candidates_per_boid'], s['occupied_cells'])
PY
BENCH_ARGS="--workload boids_2m --presteps 1000 --steady-steps 0" TAG=r04_boids_steady PMC_LAST=4 bash scripts/gpu_pmc.sh 2>&1 | tail -n 12
