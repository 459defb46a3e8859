#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  timeout -k 10 300 python bench.py --workload $1 --skip-10m --no-cpu-baseline --steps $2 --warmup 3 ${3:-} 2>/dev/null | python3 -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$1 ${3:-}', 'chunk', os.environ.get('NBMI_XCD_CHUNK'), 'fp', os.environ.get('NBMI_FORCE_PREC'), 'ms', round(d['ms_per_step'],4), 'walk', round(d['phase_ms']['walk_ms'],4))"
}
for c in 64 128 512 1024 4096; do NBMI_XCD_CHUNK=$c run collision_10m_bh 6; done
for c in 0 256; do NBMI_FORCE_PREC=1 NBMI_XCD_CHUNK=$c run collision_10m_bh 6; done
for c in 0 64 256 1024; do NBMI_XCD_CHUNK=$c run galaxy_1m_bh 8 "--bodies-per-gpu 4000000"; done
for c in 0 64 256; do NBMI_XCD_CHUNK=$c run galaxy_1m_bh 10 "--bodies-per-gpu 2000000"; done
