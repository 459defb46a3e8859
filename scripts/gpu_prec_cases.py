"""Force-precision settings of the product walk on the oracle-cached inputs of tests/oracle_cases.py (100 steps each):
max / p99.9 position error after 50 and 100 steps, float64 wave share, walk and step time.
  CASES=galaxy_1m,galaxy_1m_seed7,collision_1m,cluster_1m   MODES=auto:5e-5,auto:2e-5,f64   (auto:<tau>[:<enter>:<leave>])"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")
import numpy as np  # noqa: E402
import oracle_cases as oc  # noqa: E402
from nbody.gpu_backend import HIPBarnesHutSimulation  # noqa: E402

cases = os.environ.get("CASES", "galaxy_1m,galaxy_1m_seed7,collision_1m,cluster_1m").split(",")
modes = os.environ.get("MODES", "auto:5e-5,auto:2e-5,auto:1e-5,auto:5e-5:0.33:0.25,f64").split(",")


def make(c, p, v, m, mode):
    part = mode.split(":")
    os.environ.pop("NBMI_ALL64_ENTER", None)
    os.environ.pop("NBMI_ALL64_LEAVE", None)
    if len(part) >= 4:
        os.environ["NBMI_ALL64_ENTER"], os.environ["NBMI_ALL64_LEAVE"] = part[2], part[3]
    sim = HIPBarnesHutSimulation(p, v, m, c["G"], c["eps"], 1.0, c["theta"])
    sim.set_force_precision(part[0], float(part[1]) if len(part) > 1 else 0.0)
    return sim


for case in cases:
    c = oc.CASES[case]
    p, v, m, ref = oc.load(case, (50, 100), None, compute_if_missing=False)
    for mode in modes:
        sim = make(c, p, v, m, mode)
        row = {"case": case, "mode": mode}
        t_walk = 0.0
        for s in range(1, 101):
            sim.step(c["dt"])
            if s in (1, 50, 100):
                sh, a64 = sim.force_precision_share()
                row[f"share_{s}"] = round(sh, 3)
                row[f"all64_{s}"] = int(a64)
            if s in ref:
                e = np.abs(sim.get_positions_f64() - ref[s]).max(axis=1) / np.abs(ref[s]).max()
                row[f"max_{s}"] = float(e.max())
                row[f"p999_{s}"] = float(np.quantile(e, 0.999))
        # time of the steps 100 .. 120 (the state the error was measured in)
        sim.sync()
        sim.enable_timers(True)
        sim.timers(reset=True)
        t0 = time.perf_counter()
        sim.step_many(c["dt"], 20)
        sim.sync()
        row["ms_per_step_at_100"] = round(1e3 * (time.perf_counter() - t0) / 20, 4)
        tm = sim.timers(reset=True)
        row["walk_ms_at_100"] = round(tm["walk_ms"] / max(1, tm["steps"]), 4)
        sim.close()
        print(json.dumps(row), flush=True)
