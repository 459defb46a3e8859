#!/bin/bash
# the whole -m gpu suite, then the default bench line (what the driver runs at round end)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -s -p no:cacheprovider > gpurun_out/${TAG:-full}_pytest.log 2>&1
rc=$?
grep -E "steps:|ranks|single handle|passed|failed|skipped|rror" gpurun_out/${TAG:-full}_pytest.log | cut -c1-300 | tail -n 30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/${TAG:-full}_bench.json 2> gpurun_out/${TAG:-full}_bench_err.txt || { tail -n 20 gpurun_out/${TAG:-full}_bench_err.txt; exit 1; }
python3 - gpurun_out/${TAG:-full}_bench.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d.get('north_star_10m') or {}
print('1M', round(d['ms_per_step'],4), d['value'], d['phase_ms'], d['roofline']['frac'], d.get('config',{}).get('float64_wave_share'))
print('10M', round(t.get('ms_per_step',0),3), t.get('value'), t.get('phase_ms'), (t.get('roofline') or {}).get('frac'))
print('cpu', d.get('cpu_baseline'))
PY
