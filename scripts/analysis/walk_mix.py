"""Composition of the lock-step walk's visits: cells / leaves / leaves under a twig (a cell with only leaf children)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "3d-spatial-sim-for-boid-and-nbody_amd"))
from oracle import pyref
from tools import presets
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/walk_sim.so"
subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(here, "walk_sim.c"), "-lm"], check=True)
S = C.CDLL(so)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
np.random.seed(42)
pos, vel, m = presets.generate_distribution("galaxy", n, 800.0, 0.07)
L = pyref.lib(fast=True)
b = pyref.compute_bounds(pos, L)
nd = pyref.NodeArrays(4 * n + 64)
nn = pyref.build_octree(pos, m, b, nd, cap=pyref.UNCAPPED, L=L)
hi, lo = pyref.body_keys(pos, b, L)
order = np.lexsort((lo, hi)).astype(np.int64)
rng = np.random.default_rng(1)
gs = 64
pick = np.sort(rng.choice(n // gs, size=1500, replace=False))
sub = np.concatenate([order[g * gs:(g + 1) * gs] for g in pick]).astype(np.int64)
out = np.zeros(8, dtype=np.int64)
S.walk_mix.restype = None
S.walk_mix.argtypes = [pyref._f64p, pyref._i64p, pyref._i64, C.c_int, pyref._f64p, pyref._f64p, pyref._i32p, pyref._u8p, pyref._dbl, pyref._dbl, pyref._i64p]
S.walk_mix(pos, sub, len(sub), gs, nd.half, nd.com, nd.children, nd.leaf, 0.5, 1.5, out)
G = len(pick)
tot = out[0] + out[1]
print(f"per group: cell visits {out[0]/G:.0f} (lanes {out[3]/max(1,out[0]):.1f}), leaf visits {out[1]/G:.0f} ({100*out[1]/tot:.1f}% of visits, lanes {out[4]/max(1,out[1]):.1f}), "
      f"twig-leaf visits {out[2]/G:.0f} ({100*out[2]/tot:.1f}%, lanes {out[5]/max(1,out[2]):.1f}), twig openings {out[6]/G:.0f}, twigs among visited cells {100*out[7]/out[0]:.1f}%")
