#!/bin/bash
# per-kernel time of the owner-mode phases: the probe (W virtual ranks on one GPU) under rocprofv3 --stats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/letprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
W=${1:-8}
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 $R/scripts/gpu_let_probe.py 1000000 $W > $O/probe.jsonl 2> $O/probe.err; echo "rc=$?"
f=$(find $O/raw -name '*kernel_stats.csv' | head -1)
cp "$f" $O/let_w${W}_kernel_stats.csv
head -40 $O/let_w${W}_kernel_stats.csv | cut -c1-200
