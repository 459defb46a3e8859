#!/bin/bash
# How much of the 1 M x 100-step maximum is the luck of the summation order?  The same run with walk variants that
# evaluate the same accepted sets and differ only in the order of the fp32 sums.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/parvar
mkdir -p $O
cd $R
for v in "NBMI_WALK_PAIR=2" "NBMI_WALK_PAIR=0" "NBMI_HILBERT=0 NBMI_WALK_PAIR=2"; do
  tag=$(echo "$v" | tr ' =' '__')
  env $v N=1000000 STEPS=100 OMP_NUM_THREADS=32 timeout -k 10 600 python scripts/gpu_parity_1m.py > $O/$tag.jsonl 2> $O/$tag.err; echo "$v rc=$?"
  tail -2 $O/$tag.jsonl | head -1 | cut -c1-200
done
