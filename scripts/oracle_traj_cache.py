"""Oracle trajectory of BASELINE config 2 (galaxy, 1 M bodies, theta 0.5, dt 0.05), strict float64, saved at a few
steps so that GPU-side precision experiments (scripts/gpu_prec_diag.py) do not pay for the oracle on the GPU box.
Writes tests/cache/oracle_galaxy_<n>_step<k>.npy (git-ignored; the files travel with gpurun snapshots).
Test infrastructure: the product never reads these files."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")
import numpy as np  # noqa: E402
from oracle import pyref  # noqa: E402
from tools.presets import generate_distribution  # noqa: E402

n = int(os.environ.get("N", 1_000_000))
keep = [int(x) for x in os.environ.get("KEEP", "10,20,30,50,100").split(",")]
out = os.path.join(ROOT, "tests", "cache")
os.makedirs(out, exist_ok=True)
np.random.seed(42)
p, v, m = generate_distribution("galaxy", n, 800.0, 0.07)
cpu = pyref.BHStepper(p, v, m, 0.5, 0.07, 1.5, 1.0, cap=pyref.UNCAPPED, fast=False)
t0 = time.time()
for s in range(1, max(keep) + 1):
    cpu.step(0.05)
    if s in keep:
        np.save(os.path.join(out, f"oracle_galaxy_{n}_step{s}.npy"), cpu.pos)
        print(s, round(time.time() - t0, 1), cpu.num_nodes, flush=True)
