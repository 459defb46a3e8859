#!/bin/bash
# round 3, call E: full GPU suite + default bench (with CPU baselines)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3e_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -n 6 gpurun_out/r3e_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python bench.py > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
echo "bench rc=$?"; tail -n 3 gpurun_out/r3e_bench.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3e_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','phase_ms')})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms','lane_efficiency')})
print('cpu', d.get('cpu_baseline'))
print('frame', d.get('frame_pcie'))
n=d['north_star_10m']; print('10m', n['value'], n['ms_per_step'], n['phase_ms'], n['roofline']['frac'], n.get('cpu_baseline'))
PY
