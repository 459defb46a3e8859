"""Boids package (reference boids/__init__.py:3 exports only Flock)."""
from .flock import Flock  # noqa: F401

__all__ = ["Flock"]
