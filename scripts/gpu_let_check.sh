#!/bin/bash
# owner-mode change: its GPU tests, then the probe at W = 8 under rocprofv3 --stats and the plain probe table
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/letprof
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_sharded_record.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
bash scripts/gpu_let_prof.sh 8 > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R
timeout -k 10 500 python scripts/gpu_let_probe.py 1000000 1,2,4,8 2> $O/let_probe.err > $O/owner_mode_probe.jsonl; echo "probe rc=$?"
python3 - <<PY
import json
for l in open("$O/owner_mode_probe.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d["world"], d["rank0_ms_total"], d["rank0_ms"], d["tree_rows_received_by_rank0"])
PY
