"""MI355X-native drop-in for the hot path of Keshav-Madhav/3d-spatial-sim-for-boid-and-nbody.

The directory name is not a Python identifier; load it with
``importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")`` (repo root on sys.path) or put
this directory itself on ``sys.path``.  Either way the sub-packages are importable under the
reference's own top-level names, so code written against the reference keeps working:

    from nbody.gpu_backend import create_gpu_simulation      # reference nbody/gpu_backend.py:623
    from nbody import NBodySimulation                         # reference nbody/simulation.py:441
    from boids import Flock                                   # reference boids/flock.py:454
    from tools.record import save_frame, load_frame, record   # reference tools/record.py:88,99,702

Everything numerical runs in hand-written HIP kernels (csrc/ -> libnbmi.so) behind the C ABI of
include/nbmi.h and include/bdmi.h.  There is no CPU fallback.
"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
if _HERE not in _sys.path:
    _sys.path.insert(0, _HERE)

PACKAGE_DIR = _HERE
__all__ = ["PACKAGE_DIR"]
