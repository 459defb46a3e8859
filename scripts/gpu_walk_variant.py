"""A/B of a walk variant selected by an environment knob (e.g. NBMI_WALK_STACK=1) against the default kernel:
positions after a few steps (same accepted sets: differences are fp32 summation order) and ms per step."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")
from nbody.gpu_backend import HIPBarnesHutSimulation
from tools.presets import generate_distribution

knob, val = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
dist, R, G, eps, dt = ("galaxy", 800.0, 0.07, 1.5, 0.05) if n < 5_000_000 else ("collision", 2000.0, 0.08, 6.0, 0.25)
np.random.seed(42)
p, v, m = generate_distribution(dist, n, R, G)
base = HIPBarnesHutSimulation(p, v, m, G, eps, 1.0, 0.5)
os.environ[knob] = val
var = HIPBarnesHutSimulation(p, v, m, G, eps, 1.0, 0.5)
del os.environ[knob]
base.step_many(dt, 3); var.step_many(dt, 3); base.sync(); var.sync()
a, b = base.get_positions_f64(), var.get_positions_f64()
acc_scale = np.abs(base.accelerations()).max()
print(json.dumps({"n": n, "max_pos_diff": float(np.abs(a - b).max()), "as_acc_fraction": float(np.abs(a - b).max() / (acc_scale * dt * dt * 3))}))
for name, sim in (("default", base), (f"{knob}={val}", var)):
    sim.step_many(dt, 3); sim.sync()
    t0 = time.perf_counter(); sim.step_many(dt, 20); sim.sync(); t = (time.perf_counter() - t0) / 20
    sim.enable_timers(True); sim.timers(reset=True); sim.step_many(dt, 10); sim.sync(); tm = sim.timers(reset=True)
    print(json.dumps({"kernel": name, "ms_per_step": round(1e3 * t, 4), "walk_ms": round(tm["walk_ms"] / tm["steps"], 4),
                      "tree_ms": round(tm["tree_ms"] / tm["steps"], 4)}))
