"""theta = 0 (direct sum through the tree): which pair interactions does the four-cursor walk get wrong?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")
import numpy as np
from nbody.gpu_backend import HIPBarnesHutSimulation

rng = np.random.default_rng(3)
n = 512
pos = rng.uniform(-100, 100, (n, 3)); vel = np.zeros((n, 3)); mass = np.ones(n)
G, eps, dt = 1.0, 1.0, 0.05
os.environ["NBMI_SPLIT_WAVES"] = "0"
acc = {}
for mode in ("1", "4"):
    os.environ["NBMI_WALK_PAIR"] = mode
    s = HIPBarnesHutSimulation(pos, vel, mass, G, eps, 1.0, 0.0)
    s.build_tree(); order = s.key_order(); st = s.tree_stats()
    s.step(dt)
    acc[mode] = (s.get_positions_f64() - pos) / dt ** 2
    s.close()
rank = np.empty(n, dtype=np.int64); rank[order] = np.arange(n)
d = acc["4"] - acc["1"]
print("nodes", st["num_nodes"], "quarter nodes", st["num_nodes"] // 4)
r = pos[None, :, :] - pos[:, None, :]
d2 = (r ** 2).sum(-1) + eps ** 2
pair = G * r / d2[..., None] ** 1.5          # pair[i, j] = acceleration of i due to j
exact = pair.sum(1)
print("pair-mode error vs exact", np.abs(acc["1"] - exact).max(), " quad-mode error", np.abs(acc["4"] - exact).max())
bad = np.nonzero(np.abs(d).max(1) > 1e-4 * np.abs(exact).max())[0]
print("bodies wrong:", len(bad))
for i in bad[:40]:
    # best single j with sign
    res_m = np.linalg.norm(d[i][None, :] + pair[i], axis=1)   # missing j: d = -pair
    res_p = np.linalg.norm(d[i][None, :] - pair[i], axis=1)   # doubled j: d = +pair
    jm, jp = res_m.argmin(), res_p.argmin()
    print(f"body rank {rank[i]:4d}: |d| {np.linalg.norm(d[i]):.3e}; missing j rank {rank[jm]:4d} residual {res_m[jm]:.2e}; doubled j rank {rank[jp]:4d} residual {res_p[jp]:.2e}")
