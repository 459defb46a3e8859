"""Design analysis (CPU, oracle): what a lock-step group walk does visit by visit.

For key-sorted groups of gs bodies: histogram of taking-part lanes per group visit, share of
unanimous / mixed decisions.  Usage: python scripts/analysis/group_hist.py [n] [dist] [sample_groups]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-spatial-sim-for-boid-and-nbody_amd"))
from oracle import pyref  # noqa: E402
from tools import presets  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dist = sys.argv[2] if len(sys.argv) > 2 else "galaxy"
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
CFG = {"galaxy": (800.0, 0.07, 1.5), "collision": (2000.0, 0.08, 6.0)}
R, G, eps = CFG[dist]
theta = 0.5
np.random.seed(42)
pos, vel, m = presets.generate_distribution(dist, n, R, G)
L = pyref.lib(fast=True)
b = pyref.compute_bounds(pos, L)
nd = pyref.NodeArrays(4 * n + 64)
t0 = time.time()
nn = pyref.build_octree(pos, m, b, nd, cap=pyref.UNCAPPED, L=L)
print("nodes", nn, "build s", round(time.time() - t0, 1), flush=True)
hi, lo = pyref.body_keys(pos, b, L)
order = np.lexsort((lo, hi)).astype(np.int64)
L.nbref_group_walk_hist.restype = None
L.nbref_group_walk_hist.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, pyref._f64p, pyref._f64p,
                                    pyref._i32p, pyref._u8p, pyref._dbl, pyref._dbl, pyref._i64p, pyref._i64p]
rng = np.random.default_rng(1)
out = {}
for gs in (8, 16, 32, 64):
    ng = n // gs
    pick = np.sort(rng.choice(ng, size=min(sample, ng), replace=False))
    sub = np.concatenate([order[g * gs:(g + 1) * gs] for g in pick]).astype(np.int64)
    hist = np.zeros(gs + 1, dtype=np.int64)
    cls = np.zeros(8, dtype=np.int64)
    L.nbref_group_walk_hist(pos, sub, len(sub), gs, nd.half, nd.com, nd.children, nd.leaf, theta, eps, hist, cls)
    visits = int(hist.sum())
    lanev = int((hist * np.arange(gs + 1)).sum())
    q = np.cumsum(hist) / visits
    cum_l = np.cumsum(hist * np.arange(gs + 1)) / lanev
    row = dict(gs=gs, groups=len(pick), wave_visits_per_group=visits / len(pick),
               lane_visits_per_body=lanev / len(sub), lane_eff=lanev / (visits * gs),
               full_active_share_of_visits=float(hist[gs] / visits),
               full_active_share_of_lane_visits=float(hist[gs] * gs / lanev),
               unanimous_accept=cls[0] / visits, unanimous_open=cls[1] / visits, mixed=cls[2] / visits,
               lanes_in_unan_accept=cls[3] / lanev, lanes_in_unan_open=cls[4] / lanev, lanes_in_mixed=cls[5] / lanev,
               accepts_per_body=cls[6] / len(sub),
               visits_with_le_quarter_lanes=float(q[gs // 4]), lane_visits_with_le_quarter_lanes=float(cum_l[gs // 4]),
               visits_with_le_half_lanes=float(q[gs // 2]), lane_visits_with_le_half_lanes=float(cum_l[gs // 2]))
    out[gs] = row
    print(json.dumps(row), flush=True)
    if gs == 64:
        print("hist64 (by 8s):", [int(hist[i:i + 8].sum()) for i in range(1, 65, 8)])
