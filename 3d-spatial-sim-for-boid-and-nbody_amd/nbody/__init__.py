"""N-body package (reference nbody/__init__.py exports NBodySimulation)."""
from .simulation import NBodySimulation  # noqa: F401

__all__ = ["NBodySimulation"]
