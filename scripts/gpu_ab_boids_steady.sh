#!/bin/bash
# boids A/B: the default library against PREV="lib1.so lib2.so" (files beside it), bench line at t = 0 and after 1000 steps, two
# alternating rounds; parity suite first
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_boids.py tests/test_gpu_visibility.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 3 || exit 1
for rep in 1 2; do
for lib in default ${PREV:-}; do
  if [ "$lib" = default ]; then unset NBMI_LIB; else export NBMI_LIB=$GRAFT_REPO_ROOT/3d-spatial-sim-for-boid-and-nbody_amd/$lib; fi
  timeout -k 10 600 python bench.py --workload boids_2m --no-cpu-baseline > gpurun_out/r4h_boids_$(basename $lib).json 2> gpurun_out/r4h_err.txt || { tail -n 20 gpurun_out/r4h_err.txt; exit 1; }
  python3 - gpurun_out/r4h_boids_$(basename $lib).json $(basename $lib) <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s=d['steady_state']
print(sys.argv[2], 't=0', round(d['ms_per_step'],4), 'sweep', round(d['phase_ms']['sweep_ms'],4), '| steady', round(s['ms_per_step'],4), 'sweep', round(s['phase_ms']['sweep_ms'],4), s['candidates_per_boid'], s['occupied_cells'])
PY
done
done
