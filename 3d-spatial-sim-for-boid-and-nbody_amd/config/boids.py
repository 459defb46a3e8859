"""Boids constants, values as reference config/boids.py:30-46."""

CAMERA = {"far_clip": 1000.0}

BOIDS = {
    "count": 500000,
    "bounds": 500.0,
    "max_speed": 25.0,
    "max_force": 60.0,
    "size": 1.2,
    "wall_margin": 3.0,
    "wall_weight": 10.0,
    "perception_radius": 5.0,
    "separation_radius": 3.0,
    "separation_weight": 2.5,
    "alignment_weight": 1.0,
    "cohesion_weight": 1.0,
    "color_blend_rate": 1.0,
}

# order in which the C ABI takes them (include/bdmi.h)
PARAM_ORDER = ("bounds", "wall_margin", "wall_weight", "max_speed", "max_force", "perception_radius",
               "separation_radius", "separation_weight", "alignment_weight", "cohesion_weight",
               "color_blend_rate")
