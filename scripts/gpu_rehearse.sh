#!/bin/bash
# whole GPU suite, then the N > 1 bench paths as far as one GPU allows: two gloo ranks sharing the card, started
# by the plain command (bench.py launches its own ranks), and the one-rank RCCL path
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rehearse
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != 1 ]; then timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log; else rc=0; fi
[ $rc -ne 0 ] && exit $rc
NBMI_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/let2.json 2> $O/let2.err; echo "let gloo-2 rc=$?"; cut -c1-300 $O/let2.json
NBMI_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/let4.json 2> $O/let4.err; echo "let gloo-4 rc=$?"; cut -c1-300 $O/let4.json
NBMI_BENCH_BACKEND=gloo NBMI_SHARD_MODE=rows timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/rows2.json 2> $O/rows2.err; echo "rows gloo-2 rc=$?"; cut -c1-200 $O/rows2.json
NBMI_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload boids_2m --steps 5 --warmup 2 --no-cpu-baseline > $O/boids2.json 2> $O/boids2.err; echo "boids gloo-2 rc=$?"; cut -c1-200 $O/boids2.json
NBMI_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rccl1.json 2> $O/rccl1.err; echo "rccl-1 rc=$?"; cut -c1-200 $O/rccl1.json
