#!/bin/bash
# the whole GPU suite, then the default bench line (1 M + 10 M) without the CPU baseline, twice
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/full
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
fi
for rep in 1 2; do
  for hil in 1 0; do
    NBMI_HILBERT=$hil timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
    python3 - "$O/bench.json" "$hil" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d.get("north_star_10m") or {}
print("hilbert", sys.argv[2], "1M", round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phase_ms"].items()}, "10M", round(t.get("ms_per_step", 0), 3), {k: round(v, 3) for k, v in t.get("phase_ms", {}).items()})
PY
  done
done
