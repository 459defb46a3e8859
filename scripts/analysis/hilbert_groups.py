"""Would waves formed along a Hilbert curve walk fewer nodes than waves formed along the octant (Morton) order?
Same octree, same opening tests; only which 64 bodies share a wave changes.  CPU, oracle tree.
python scripts/analysis/hilbert_groups.py [n] [dist] [groups]"""
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


def hilbert_index(ix, iy, iz, bits):
    """Skilling's transpose algorithm, vectorised: integer coordinates -> Hilbert index (3 * bits bits)."""
    X = [ix.astype(np.uint64).copy(), iy.astype(np.uint64).copy(), iz.astype(np.uint64).copy()]
    M = np.uint64(1) << np.uint64(bits - 1)
    Q = M
    while Q > 1:
        P = Q - np.uint64(1)
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            X[0] = np.where(m, X[0], X[0] ^ t)
            X[i] = np.where(m, X[i], X[i] ^ t)
        Q >>= np.uint64(1)
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - np.uint64(1)), t)
        Q >>= np.uint64(1)
    for i in range(3):
        X[i] ^= t
    h = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            h = (h << np.uint64(1)) | ((X[i] >> np.uint64(b)) & np.uint64(1))
    return h


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
morton = np.lexsort((lo, hi)).astype(np.int64)
bits = 20
q = np.clip(((pos + b) / (2 * b) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
hil = np.argsort(hilbert_index(q[:, 0], q[:, 1], q[:, 2], bits), kind="stable").astype(np.int64)
rank = np.zeros(nn, dtype=np.int32)
S.preorder_rank.restype = None
S.preorder_rank.argtypes = [pyref._i32p, pyref._u8p, pyref._i64, pyref._i32p]
S.preorder_rank(nd.children, nd.leaf, nn, rank)
S.range_visits.restype = None
S.range_visits.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, pyref._f64p, pyref._f64p, pyref._i32p, pyref._u8p,
                           pyref._i32p, pyref._i64, pyref._dbl, pyref._dbl, C.c_int, pyref._i64p]
gs, NB = 64, 4
rng = np.random.default_rng(1)
pick = np.sort(rng.choice(n // gs, size=min(sample, n // gs), replace=False))
for name, order in (("octant (Morton) order", morton), ("Hilbert order", hil)):
    sub = np.concatenate([order[g * gs:(g + 1) * gs] for g in pick]).astype(np.int64)
    out = np.zeros((len(pick), NB), dtype=np.int64)
    S.range_visits(pos, sub, len(sub), gs, nd.half, nd.com, nd.children, nd.leaf, rank, nn, 0.5, eps, NB, out)
    print(f"{dist} n={n} {name}: wave-visits per group {out.sum(1).mean():.0f}")
