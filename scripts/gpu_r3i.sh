#!/bin/bash
# walk block size / XCD chunk sweep with the mixed fp32 / float64 waves (auto precision)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --workload $1 --skip-10m --no-cpu-baseline --steps $2 --warmup 3 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$1', 'chunk', os.environ.get('NBMI_XCD_CHUNK'), 'block', os.environ.get('NBMI_WALK_BLOCK'), 'ms', round(d['ms_per_step'],4), 'walk', round(d['phase_ms']['walk_ms'],4))"
}
for c in 0 4 16 64 256 -1; do NBMI_XCD_CHUNK=$c run galaxy_1m_bh 20; done
for b in 128 64; do NBMI_WALK_BLOCK=$b run galaxy_1m_bh 20; done
for c in 0 16 256; do NBMI_XCD_CHUNK=$c run collision_10m_bh 6; done
NBMI_WALK_BLOCK=64 run collision_10m_bh 6
