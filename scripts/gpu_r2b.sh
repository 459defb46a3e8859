#!/bin/bash
# quick check of a walk change: the N-body GPU tests (accepted sets, shard bit-identity, long runs), then the
# default bench line (1 M + 10 M) without the CPU baseline
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_nbody.py tests/test_gpu_sharded_record.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["phase_ms"], d["value"])
t=d.get("north_star_10m")
if t: print(t["ms_per_step"], t["phase_ms"], t["value"])
PY
