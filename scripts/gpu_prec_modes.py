"""Force-precision modes of the product walk at BASELINE config 2 (galaxy, 1 M bodies, 100 steps against the cached
oracle trajectory) and their price: max / p99.9 position error, walk and step time per mode.  MODES = list of
f32 | f64 | auto:<tau>.  With N10M=1 also the timing of the 10 M-body collision (no oracle at that size)."""
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
from tools.presets import generate_distribution  # noqa: E402

modes = os.environ.get("MODES", "f32,auto:5e-5,auto:2e-5,auto:1e-5,auto:5e-6,f64").split(",")


def setup(sim, mode):
    md, _, tau = mode.partition(":")
    sim.set_force_precision(md, float(tau) if tau else 0.0)


def timing(sim, dt, steps=20):
    sim.step_many(dt, 3)
    sim.sync()
    t0 = time.perf_counter()
    sim.step_many(dt, steps)
    sim.sync()
    el = (time.perf_counter() - t0) / steps
    sim.enable_timers(True)
    sim.timers(reset=True)
    sim.step_many(dt, steps)
    sim.sync()
    tm = sim.timers(reset=True)
    sim.enable_timers(False)
    k = max(1, tm["steps"])
    return {"ms_per_step": 1e3 * el, "walk_ms": tm["walk_ms"] / k, "tree_ms": tm["tree_ms"] / k, "sort_ms": tm["sort_ms"] / k,
            "keys_ms": tm["keys_ms"] / k}


n = 1_000_000
np.random.seed(42)
p, v, m = generate_distribution("galaxy", n, 800.0, 0.07)
ref = {k: np.load(os.path.join(ROOT, "tests", "cache", f"oracle_galaxy_{n}_step{k}.npy")) for k in (50, 100)}
for mode in modes:
    sim = HIPBarnesHutSimulation(p, v, m, 0.07, 1.5, 1.0, 0.5)
    setup(sim, mode)
    row = {"workload": "galaxy_1m", "mode": mode}
    for s in range(1, 101):
        sim.step(0.05)
        if s in ref:
            e = np.abs(sim.get_positions_f64() - ref[s]).max(axis=1) / np.abs(ref[s]).max()
            row[f"max_{s}"] = float(e.max())
            row[f"p999_{s}"] = float(np.quantile(e, 0.999))
            row[f"n_over_1e-5_{s}"] = int((e > 1e-5).sum())
    sim.close()
    sim = HIPBarnesHutSimulation(p, v, m, 0.07, 1.5, 1.0, 0.5)
    setup(sim, mode)
    row.update(timing(sim, 0.05))
    sim.close()
    print(json.dumps(row), flush=True)
if os.environ.get("N10M"):
    np.random.seed(42)
    p, v, m = generate_distribution("collision", 10_000_000, 2000.0, 0.08)
    for mode in os.environ.get("MODES10M", "f32,auto:1e-5,f64").split(","):
        sim = HIPBarnesHutSimulation(p, v, m, 0.08, 6.0, 1.0, 0.5)
        setup(sim, mode)
        row = {"workload": "collision_10m", "mode": mode}
        row.update(timing(sim, 0.25, steps=10))
        sim.close()
        print(json.dumps(row), flush=True)
