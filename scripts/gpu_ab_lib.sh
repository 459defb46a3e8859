#!/bin/bash
# A/B of several builds of the library on one box: the default libnbmi.so vs the files named on the command
# line (paths relative to the repo, selected with NBMI_LIB); two alternating rounds
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset NBMI_LIB; else export NBMI_LIB=$R/$lib; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_${rep}_$(basename $lib).json 2> $O/err.txt || { echo "bench failed"; tail -5 $O/err.txt; exit 1; }
    python3 - "$O/bench_${rep}_$(basename $lib).json" "$(basename $lib)" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d.get("north_star_10m") or {}
print(sys.argv[2], "1M", round(d["ms_per_step"], 4), round(d["phase_ms"]["walk_ms"], 4), "10M", round(t.get("ms_per_step", 0), 3), round(t.get("phase_ms", {}).get("walk_ms", 0), 3))
PY
  done
done
