"""BASELINE's accuracy criterion at the bench size: galaxy, 1 M bodies, theta 0.5, dt 0.05, 100 steps on the
GPU against the CPU oracle (strict IEEE build of oracle/nbref.c = the reference's algorithm in float64,
all host threads).  Prints one JSON line per 10 steps (also keeps the run from looking hung) and a
summary; the oracle needs ~3 s per step, so this is a one-off measurement, not a test."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")
import numpy as np  # noqa: E402
from nbody.gpu_backend import HIPBarnesHutSimulation  # noqa: E402
from oracle import pyref  # noqa: E402
from tools.presets import generate_distribution  # noqa: E402

n = int(os.environ.get("N", 1_000_000))
steps = int(os.environ.get("STEPS", 100))
np.random.seed(42)
p, v, m = generate_distribution("galaxy", n, 800.0, 0.07)
gpu = HIPBarnesHutSimulation(p, v, m, 0.07, 1.5, 1.0, 0.5)
cpu = pyref.BHStepper(p, v, m, 0.5, 0.07, 1.5, 1.0, cap=pyref.UNCAPPED, fast=False)
t0 = time.time()
out = []
for s in range(1, steps + 1):
    gpu.step(0.05)
    cpu.step(0.05)
    if s % 10 == 0 or s == 1:
        gp = gpu.get_positions_f64()
        err = np.abs(gp - cpu.pos)
        scale = np.abs(cpu.pos).max()
        row = {"step": s, "max_rel_pos_err": float(err.max() / scale), "rms_rel_pos_err": float(np.sqrt((err ** 2).mean()) / scale),
               "p999_rel": float(np.quantile(err.max(axis=1), 0.999) / scale), "nodes_gpu": gpu.tree_stats()["num_nodes"],
               "nodes_cpu": cpu.num_nodes, "elapsed_s": round(time.time() - t0, 1)}
        out.append(row)
        print(json.dumps(row), flush=True)
# who holds the maximum?  the worst bodies, their error in length units and the distance to their nearest neighbour
try:
    from scipy.spatial import cKDTree
    gp = gpu.get_positions_f64()
    e = np.linalg.norm(gp - cpu.pos, axis=1)
    worst = np.argsort(e)[-8:][::-1]
    dist, idx = cKDTree(cpu.pos).query(cpu.pos[worst], k=2)
    print(json.dumps({"worst_bodies": [{"body": int(b), "abs_err": float(e[b]), "nearest_neighbour_dist": float(d[1]),
                                        "err_over_neighbour_dist": float(e[b] / d[1]), "neighbour": int(i[1]),
                                        "neighbour_abs_err": float(e[i[1]])} for b, d, i in zip(worst, dist, idx)],
                      "softening": 1.5, "largest_coordinate": float(np.abs(cpu.pos).max())}))
except Exception as ex:  # diagnostics only
    print(json.dumps({"worst_bodies_error": repr(ex)}))
print(json.dumps({"summary": {"n": n, "steps": steps, "theta": 0.5, "dt": 0.05, "oracle": "strict float64, uncapped",
                              "max_rel_pos_err_final": out[-1]["max_rel_pos_err"], "threads": int(pyref.lib().nbref_num_threads())}}))
