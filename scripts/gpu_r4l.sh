#!/bin/bash
# where a small system's step goes: kernel trace of galaxy_10k_bh (config 1's size)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_10k -- python3 $GRAFT_REPO_ROOT/bench.py --workload galaxy_10k_bh --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r4l_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r4l_err.txt
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections
d=json.loads(open('gpurun_out/r4l_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], d.get('phase_ms'))
f=glob.glob('gpurun_out/prof_10k/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# one step in the middle: find consecutive k_keys
idx=[i for i,r in enumerate(rows) if 'k_keys' in r['Kernel_Name']]
a,b=idx[len(idx)//2], idx[len(idx)//2+1]
t0=int(rows[a]['Start_Timestamp'])
busy=0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    busy+=e-s
    print(f"{(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  {r['Kernel_Name'][:70]}")
print('step span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3, 'kernel busy us', busy/1e3, 'launches', b-a)
PY
