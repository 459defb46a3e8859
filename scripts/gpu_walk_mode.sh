#!/bin/bash
# a walk variant selected by NBMI_WALK_PAIR (2 = two cursors cut at the wave's own leaves) vs the default: parity tests with it, then alternating bench runs
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quad
mkdir -p $O
cd $R
NBMI_WALK_PAIR=2 timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest(home split) rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for mode in 1 2; do
    NBMI_WALK_PAIR=$mode timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_${rep}_$mode.json 2> $O/err.txt || { echo "bench failed"; tail -5 $O/err.txt; exit 1; }
    python3 - "$O/bench_${rep}_$mode.json" "$mode" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d.get("north_star_10m") or {}
print("cursors", sys.argv[2], "1M", round(d["ms_per_step"], 4), round(d["phase_ms"]["walk_ms"], 4), "10M", round(t.get("ms_per_step", 0), 3), round(t.get("phase_ms", {}).get("walk_ms", 0), 3))
PY
  done
done
