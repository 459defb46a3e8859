#!/bin/bash
# boids_2m with several builds of the library (NBMI_LIB), two alternating rounds
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/abb
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset NBMI_LIB; else export NBMI_LIB=$R/$lib; fi
    timeout -k 10 300 python bench.py --workload boids_2m --no-cpu-baseline --steps 30 --warmup 3 > $O/b.json 2> $O/err.txt || { echo "bench failed"; tail -5 $O/err.txt; exit 1; }
    python3 - "$O/b.json" "$(basename $lib)" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"], 4), d.get("phase_ms"), d["roofline"].get("kernel_ms"))
PY
  done
done
