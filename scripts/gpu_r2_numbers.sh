#!/bin/bash
# Round-2 evidence, part 2: the bench lines (default with the 10 M object and CPU baselines; the other configs),
# the 100-step parity run at 1 M bodies, the owner-mode probe.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/r02_default_bench.json 2> $O/default_bench.err; echo "default bench rc=$?"
cut -c1-400 $O/r02_default_bench.json
for w in boids_2m cluster_1m_direct; do
  steps=20; [ "$w" = cluster_1m_direct ] && steps=3
  timeout -k 10 400 python bench.py --workload $w --steps $steps --warmup 1 2> $O/cfg_$w.err >> $O/r02_other_configs_bench.jsonl; echo "$w rc=$?"
done
for th in 0.8 0.95 1.3; do
  timeout -k 10 200 python bench.py --theta $th --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null >> $O/r02_theta_sweep_bench.jsonl; echo "theta $th rc=$?"
done
N=1000000 STEPS=100 OMP_NUM_THREADS=32 timeout -k 10 900 python scripts/gpu_parity_1m.py > $O/r02_parity_galaxy_1m_100steps.jsonl 2> $O/parity.err; echo "parity rc=$?"
tail -2 $O/r02_parity_galaxy_1m_100steps.jsonl
timeout -k 10 500 python scripts/gpu_let_probe.py 1000000 1,2,4,8 2> $O/let_probe.err > $O/r02_owner_mode_probe.jsonl; echo "probe rc=$?"
PROBE_SCALE_RADIUS=0 timeout -k 10 300 python scripts/gpu_let_probe.py 1000000 8 2> $O/let_probe_dense.err > $O/r02_owner_mode_probe_bench_radius.jsonl; echo "probe (bench radius) rc=$?"
