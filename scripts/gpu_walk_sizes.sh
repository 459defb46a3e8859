#!/bin/bash
# where the home cut (NBMI_WALK_PAIR=2) and the model cut (=3) overtake the middle cut (=1): walk ms by size
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sizes
mkdir -p $O
cd $R
for wl in galaxy_1m_bh collision_10m_bh; do
  for nb in 1000000 2000000 4000000 10000000; do
    for mode in 1 2 3; do
      NBMI_WALK_PAIR=$mode timeout -k 10 300 python bench.py --workload $wl --bodies-per-gpu $nb --skip-10m --no-cpu-baseline --steps 10 --warmup 2 > $O/b.json 2> $O/err.txt || { echo "bench failed"; tail -5 $O/err.txt; exit 1; }
      python3 - "$O/b.json" "$wl" "$nb" "$mode" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "mode", sys.argv[4], "step", round(d["ms_per_step"], 4), "walk", round(d["phase_ms"]["walk_ms"], 4))
PY
    done
  done
done
