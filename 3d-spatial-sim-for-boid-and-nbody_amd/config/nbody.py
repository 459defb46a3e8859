"""N-body constants, values as reference config/nbody.py:16-17, 57-73 (module-level dicts are
the reference's whole "flag system"; edit here to change the live simulation)."""

BODY_COUNT = 150_000  # reference "MEDIUM" preset
THETA = 0.8

CAMERA = {"far_clip": 5000.0}

NBODY = {
    "count": BODY_COUNT,
    "spawn_radius": 500.0,
    "G": 0.1,
    "theta": THETA,
    "softening": 2.0,
    "damping": 1.0,
    "distribution": "galaxy",  # galaxy | spiral | sphere | collision | uniform
    "point_size": 1.5,
    "max_speed_color": 15.0,
}
