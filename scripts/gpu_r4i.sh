#!/bin/bash
# how does the boids sweep depend on the waves per SIMD?  (LDS padding lowers the occupancy from 4 to 3 and 2)
set -u
mkdir -p gpurun_out
for pad in 0 8000 20000 48000; do
  BDMI_LDS_PAD=$pad timeout -k 10 600 python bench.py --workload boids_2m --no-cpu-baseline > gpurun_out/r4i.json 2> gpurun_out/r4i_err.txt || { tail -n 20 gpurun_out/r4i_err.txt; exit 1; }
  python3 - $pad <<'PY'
import json, sys
d=json.loads(open('gpurun_out/r4i.json').read().strip().splitlines()[-1])
s=d['steady_state']
print('LDS pad', sys.argv[1], 't=0 sweep', round(d['phase_ms']['sweep_ms'],4), '| steady sweep', round(s['phase_ms']['sweep_ms'],4))
PY
done
