#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes, kernel-trace only).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG:-x}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/counters_list.txt" 2>&1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "=== pass $i: $line"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d "$out/pass$i" -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/pass$i.log" 2>&1
  rc=$?
  echo "rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
done <<PASSES
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
PASSES
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_'+os.environ.get('TAG','x')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/pass*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for k,d in agg.items():
        if 'k_walk' in k or 'k_emit' in k or 'k_direct' in k or 'k_flock' in k:
            fo.write(k+'\n')
            for c,v in sorted(d.items()):
                fo.write(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out+'/summary.txt').read())
PY
