"""Drives range_balance.c: how a wave's visits spread over the pre-order node array, and what that means for
K cursors on K equal parts of it (time = visits while all K run, then the rest with fewer).
python scripts/analysis/range_balance.py [n] [dist] [groups]"""
import ctypes as C
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
so = "/tmp/range_balance.so"
subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "range_balance.c"), "-lm"], check=True)
S = C.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dist = sys.argv[2] if len(sys.argv) > 2 else "galaxy"
sample = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
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
rank = np.zeros(nn, dtype=np.int32)
S.preorder_rank.restype = None
S.preorder_rank.argtypes = [pyref._i32p, pyref._u8p, pyref._i64, pyref._i32p]
S.preorder_rank(nd.children, nd.leaf, nn, rank)
S.range_visits.restype = None
S.range_visits.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, pyref._f64p, pyref._f64p, pyref._i32p, pyref._u8p,
                           pyref._i32p, pyref._i64, pyref._dbl, pyref._dbl, C.c_int, pyref._i64p]
gs, NB = 64, 64
rng = np.random.default_rng(1)
ng = n // gs
pick = np.sort(rng.choice(ng, size=min(sample, ng), replace=False))
sub = np.concatenate([order[g * gs:(g + 1) * gs] for g in pick]).astype(np.int64)
out = np.zeros((len(pick), NB), dtype=np.int64)
S.range_visits(pos, sub, len(sub), gs, nd.half, nd.com, nd.children, nd.leaf, rank, nn, 0.5, eps, NB, out)
tot = out.sum(1)
print(f"{dist} n={n}: nodes {nn}, groups {len(pick)}, wave-visits per group {tot.mean():.0f}")
home = out.argmax(1)
print(f"share of a group's visits in its busiest 1/64 of the array: {np.mean(out.max(1) / tot):.3f}; busiest 1/8: "
      f"{np.mean(out.reshape(len(pick), 8, 8).sum(2).max(1) / tot):.3f}")
for K in (2, 4, 8):
    parts = out.reshape(len(pick), K, NB // K).sum(2)          # visits of each of the K equal parts
    srt = np.sort(parts, axis=1)                                  # ascending
    # phases: all K cursors run until the shortest part ends, then K-1, ...: trips of phase j = srt[j] - srt[j-1]
    trips = np.diff(np.concatenate([np.zeros((len(pick), 1), dtype=np.int64), srt], axis=1), axis=1)
    live = K - np.arange(K)                                       # cursors alive in each phase
    print(f"K={K}: share of visits done while all {K} cursors run {np.mean(K * srt[:, 0] / tot):.3f}; mean cursors alive per "
          f"visit {np.mean((trips * live * live).sum(1) / tot):.2f}; trips / visits {np.mean(trips.sum(1) / tot):.3f} (ideal {1 / K:.3f})")


# ---- splits placed at the group's own position in the array ------------------------------------------------
CAP = 16384
S.visit_ranks.restype = None
S.visit_ranks.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, pyref._f64p, pyref._f64p, pyref._i32p, pyref._u8p,
                          pyref._i32p, pyref._dbl, pyref._dbl, pyref._i64, pyref._i32p, pyref._i64p]
vr = np.zeros((len(pick), CAP), dtype=np.int32)
cnt = np.zeros(len(pick), dtype=np.int64)
S.visit_ranks(pos, sub, len(sub), gs, nd.half, nd.com, nd.children, nd.leaf, rank, 0.5, eps, CAP, vr, cnt)
leaf_of_body = np.full(n, -1, dtype=np.int32)
lb = np.nonzero((nd.leaf[:nn] == 1) & (nd.body[:nn] >= 0))[0]
leaf_of_body[nd.body[lb]] = lb
assert (leaf_of_body >= 0).all()
def coverage(splits_fn, K, label):
    """splits_fn(g) -> K-1 ascending pre-order ranks; reports trips / visits for K cursors"""
    tv, full = [], []
    for gi in range(len(pick)):
        r = vr[gi, :cnt[gi]]
        sp = splits_fn(gi)
        parts = np.bincount(np.searchsorted(sp, r, side="right"), minlength=K)
        srt = np.sort(parts)
        tv.append(srt[-1] / cnt[gi])          # trips if every phase runs with all remaining cursors in lock-step
        full.append(K * srt[0] / cnt[gi])
    print(f"{label}: trips / visits {np.mean(tv):.3f} (ideal {1 / K:.3f}); share with all {K} cursors {np.mean(full):.3f}")
mid_leaf = lambda gi: rank[leaf_of_body[sub[gi * gs + gs // 2]]]
first_leaf = lambda gi: rank[leaf_of_body[sub[gi * gs]]]
last_leaf = lambda gi: rank[leaf_of_body[sub[gi * gs + gs - 1]]]
coverage(lambda gi: np.array([nn // 2]), 2, "K=2, equal halves")
coverage(lambda gi: np.array([mid_leaf(gi)]), 2, "K=2, split at the leaf of the group's middle body")
for d in (500, 1000, 2000, 4000, 8000, 16000):
    coverage(lambda gi: np.array([max(0, first_leaf(gi) - d), mid_leaf(gi), min(nn - 1, last_leaf(gi) + d)]), 4,
             f"K=4, splits at first leaf - {d}, middle leaf, last leaf + {d}")

# ---- model-based split: far field uniform over the array (share 1 - h), home share h around the own leaves ---
def model_split(gi, h):
    H0, H1 = first_leaf(gi), last_leaf(gi)
    x = 0.5 * (H0 + H1) / nn
    far = 1.0 - h
    if far * x + h < 0.5:      # even with all of the home on the left the left part is too light: go right of home
        return np.array([max(H1 + 1, int((0.5 - h) / far * nn))])
    if far * x > 0.5:          # left of home
        return np.array([min(H0, int(0.5 / far * nn))])
    k = int(round((gs - 1) * (0.5 - far * x) / h))
    return np.array([rank[leaf_of_body[sub[gi * gs + min(gs - 1, max(0, k))]]]])
for h in (0.3, 0.42, 0.5, 0.6):
    coverage(lambda gi: model_split(gi, h), 2, f"K=2, model split, home share {h}")
# the best any single split could do (per-group median of the visit ranks)
coverage(lambda gi: np.array([int(np.median(vr[gi, :cnt[gi]]))]), 2, "K=2, split at the median visit (oracle)")
coverage(lambda gi: np.quantile(vr[gi, :cnt[gi]], [0.25, 0.5, 0.75]).astype(np.int64), 4, "K=4, splits at the visit quartiles (oracle)")
