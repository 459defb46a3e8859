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
import csv, glob, os, collections, json
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_'+os.environ.get('TAG','x')
KEYS=('k_walk<true','k_walk<false','k_emit_tile','k_gather_scan','k_scan_subtiles','k_keys','k_tiefix','k_radix_pass','k_radix_hist','k_maxabs','k_direct','k_flock<true','k_reorder','k_table','k_assign')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
LAST=int(os.environ.get('PMC_LAST','0'))  # > 0: only the last PMC_LAST dispatches of each kernel (a state reached late in the run)
for f in glob.glob(out+'/pass*/*/*counter_collection.csv'):
    seen=set()
    rows=list(csv.DictReader(open(f)))
    if LAST:
        ids=collections.defaultdict(set)
        for r in rows:
            k=[x for x in KEYS if x in r['Kernel_Name']]
            if k: ids[k[0]].add(int(r['Dispatch_Id']))
        keep={k: set(sorted(v)[-LAST:]) for k,v in ids.items()}
        rows=[r for r in rows if any(x in r['Kernel_Name'] and int(r['Dispatch_Id']) in keep[x] for x in keep)]
    for r in rows:
        n=r['Kernel_Name']
        k=[x for x in KEYS if x in n]
        if not k: continue
        agg[k[0]][r['Counter_Name']].append(float(r['Counter_Value']))
        did=(f, r['Dispatch_Id'])
        if did not in seen:
            seen.add(did); dur[k[0]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
res={}
for k,d in agg.items():
    res[k]={c: sum(v)/len(v) for c,v in d.items()}
    res[k]['_dispatches']=max(len(v) for v in d.values())
    res[k]['_avg_us_under_pmc']=sum(dur[k])/len(dur[k])
    fs=res[k].get('FETCH_SIZE'); ws=res[k].get('WRITE_SIZE')
    if fs is not None and ws is not None:
        # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE reads 1/2 for wide coalesced reads
        res[k]['hbm_bytes_raw']=(fs+ws)*1024
        res[k]['hbm_bytes_fetch_x2']=(2*fs+ws)*1024
import hashlib
h=hashlib.sha256()
d=os.environ.get('GRAFT_REPO_ROOT','.')+'/3d-spatial-sim-for-boid-and-nbody_amd/csrc'
for name in sorted(os.listdir(d)):
    if name.endswith(('.hip','.h')):
        h.update(name.encode()+b'\0'+open(os.path.join(d,name),'rb').read())
res['_csrc_sha']=h.hexdigest()[:16]   # bench.py only quotes this summary for the kernels it was taken from
json.dump(res, open(out+'/summary.json','w'), indent=1, sort_keys=True)
for k,d in res.items():
    if not isinstance(d, dict): continue
    print(k, {c: (round(v,1) if isinstance(v,float) else v) for c,v in d.items() if c in ('FETCH_SIZE','WRITE_SIZE','hbm_bytes_raw','hbm_bytes_fetch_x2','_avg_us_under_pmc','TCC_HIT_sum','TCC_MISS_sum','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_WAVES','SQ_WAIT_ANY','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS','SQ_BUSY_CYCLES')})
PY
