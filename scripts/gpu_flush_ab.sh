#!/bin/bash
# flush cadence of the two-level sums: in-tree build vs $1 - one-step position error at 1 M bodies and the bench
set -u
R=$GRAFT_REPO_ROOT
cd $R
for lib in default "$1"; do
  if [ "$lib" = default ]; then unset NBMI_LIB; else export NBMI_LIB=$R/$lib; fi
  echo "== $lib"
  N=1000000 STEPS=10 OMP_NUM_THREADS=32 timeout -k 10 300 python scripts/gpu_parity_1m.py 2>/dev/null | grep '"step"' | cut -c1-140
done
unset NBMI_LIB
bash scripts/gpu_ab_lib.sh "$1"
