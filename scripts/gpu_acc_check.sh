#!/bin/bash
# two-level sums: whole GPU suite, default bench (no CPU baseline), then the 1 M x 100-step parity run
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/acc
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python3 - "$O/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d.get("north_star_10m") or {}
print("1M", round(d["ms_per_step"], 4), {k: round(v, 4) for k, v in d["phase_ms"].items()}, "10M", round(t.get("ms_per_step", 0), 3), {k: round(v, 3) for k, v in t.get("phase_ms", {}).items()})
PY
N=1000000 STEPS=100 OMP_NUM_THREADS=32 timeout -k 10 600 python scripts/gpu_parity_1m.py > $O/parity.jsonl 2> $O/parity.err; echo "parity rc=$?"
grep '"step"' $O/parity.jsonl | cut -c1-170
