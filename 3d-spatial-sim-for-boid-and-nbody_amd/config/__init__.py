"""Constants of the reference's config package (config/nbody.py, config/boids.py)."""
from . import boids, nbody  # noqa: F401
