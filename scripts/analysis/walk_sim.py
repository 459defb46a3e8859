"""Drives walk_sim.c: lock-step walk with deferral threshold T.  python scripts/analysis/walk_sim.py [n] [dist] [groups]"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-spatial-sim-for-boid-and-nbody_amd"))
from oracle import pyref  # noqa: E402
from tools import presets  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/walk_sim.so"
subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "walk_sim.c"), "-lm"], check=True)
S = C.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dist = sys.argv[2] if len(sys.argv) > 2 else "galaxy"
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
CFG = {"galaxy": (800.0, 0.07, 1.5), "collision": (2000.0, 0.08, 6.0)}
R, G, eps = CFG[dist]
np.random.seed(42)
pos, vel, m = presets.generate_distribution(dist, n, R, G)
L = pyref.lib(fast=True)
b = pyref.compute_bounds(pos, L)
nd = pyref.NodeArrays(4 * n + 64)
nn = pyref.build_octree(pos, m, b, nd, cap=pyref.UNCAPPED, L=L)
hi, lo = pyref.body_keys(pos, b, L)
order = np.lexsort((lo, hi)).astype(np.int64)
S.walk_sim.restype = None
S.walk_sim.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, C.c_int, C.c_int, pyref._f64p, pyref._f64p,
                       pyref._i32p, pyref._u8p, pyref._dbl, pyref._dbl, pyref._i64p]
rng = np.random.default_rng(1)
gs = 64
ng = n // gs
pick = np.sort(rng.choice(ng, size=min(sample, ng), replace=False))
sub = np.concatenate([order[g * gs:(g + 1) * gs] for g in pick]).astype(np.int64)
for T in (0, 2, 4, 8, 12, 16, 24, 32, 48):
    for fq in (64, 128):
        if T == 0 and fq != 64:
            continue
        out = np.zeros(8, dtype=np.int64)
        S.walk_sim(pos, sub, len(sub), gs, T, fq, nd.half, nd.com, nd.children, nd.leaf, 0.5, eps, out)
        G_ = len(pick)
        print(json.dumps(dict(T=T, flushq=fq, lock_visits=out[0] / G_, lock_eff=out[1] / (out[0] * gs), events=out[2] / G_,
                              tasks=out[3] / G_, task_lane_visits_per_body=out[4] / len(sub), mean_task=out[4] / max(1, out[3]),
                              drain_iters=out[5] / G_, drains=out[6] / G_, drain_eff=out[4] / max(1, out[5] * gs),
                              max_task=int(out[7]))), flush=True)
