"""Initial-condition generators and presets needed by the hot path's harness.

Restates the three distributions BASELINE.json names - ``galaxy``, ``collision``, ``cluster`` -
of the reference's generate_distribution (tools/presets.py:91-232, :350-397) and its rotation
curve helper (:52-88), drawing from the GLOBAL NumPy RNG in the same order with the same
float64 expressions, so ``np.random.seed(s)`` before the call reproduces the reference's arrays
bit for bit (pinned by tests/golden/ic_pins.npz).  The other 22 distributions and 61 presets of
the reference are content, not compute, and are not provided.
"""
from typing import Tuple

import numpy as np

DISTRIBUTIONS = {
    "galaxy": "Classic spiral disk galaxy",
    "collision": "Two galaxies colliding",
    "cluster": "Dense star cluster (globular)",
}


def compute_rotation_curve(r: np.ndarray, masses: np.ndarray, G: float, softening: float) -> np.ndarray:
    """Circular speed of a softened disk from the enclosed (radius-sorted) mass; reference :52-88."""
    order = np.argsort(r)
    rs = r[order]
    enclosed = np.cumsum(masses[order])
    eps = softening * 2
    eps_sq = eps ** 2
    r_sq = rs ** 2
    v = np.sqrt(G * enclosed * r_sq / (r_sq + eps_sq) ** 1.5)
    inner_scale = softening * 2
    v *= np.maximum((rs ** 2) / (rs ** 2 + inner_scale ** 2), 0.3)
    return v[np.argsort(order)]


def _soft_truncated_radii(count, scale_length, max_r, floor):
    """Exp(scale) radii, softly capped near max_r, floored (reference :110-116, :161-165)."""
    r = np.random.exponential(scale_length, count)
    r = r * (1 - np.exp(-max_r / (r + 0.01)))
    return np.maximum(r, floor)


def _disk_galaxy(pos, vel, masses, R, G, count, scale_length, softening, max_r, height, disp, spin,
                 x0=0.0, y0=0.0):
    """One rotating exponential disk written into the (count,3) views pos/vel."""
    r = _soft_truncated_radii(count, scale_length, max_r, R * 0.001)
    theta = np.random.uniform(0, 2 * np.pi, count)
    if x0 == 0.0 and y0 == 0.0:
        disk_height = R * height * (1 + (r / R) ** 0.5 * 0.3)
        z = np.random.normal(0, 1, count) * disk_height
        pos[:, 0] = r * np.cos(theta)
        pos[:, 1] = z
        pos[:, 2] = r * np.sin(theta)
    else:
        pos[:, 0] = r * np.cos(theta) + x0
        disk_height = R * height * (1 + (r / R) ** 0.5 * 0.3)
        if y0 == 0.0:
            pos[:, 1] = np.random.normal(0, 1, count) * disk_height
        else:
            pos[:, 1] = np.random.normal(0, 1, count) * disk_height + y0
        pos[:, 2] = r * np.sin(theta)
    speed = compute_rotation_curve(r, masses, G, softening)
    if spin > 0:  # counter-clockwise in the XZ plane
        vel[:, 0] = -speed * np.sin(theta)
        vel[:, 2] = speed * np.cos(theta)
    else:  # clockwise (second galaxy of "collision")
        vel[:, 0] = speed * np.sin(theta)
        vel[:, 2] = -speed * np.cos(theta)
    radial_factor = r / (r + softening * 2)
    sigma = speed * disp * radial_factor + np.sqrt(G * count * 0.00005)
    vel[:, 0] += np.random.normal(0, sigma, count)
    vel[:, 2] += np.random.normal(0, sigma, count)
    vel[:, 1] = np.random.normal(0, sigma * 0.25, count)


def generate_distribution(distribution: str, n: int, R: float, G: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(positions (n,3), velocities (n,3), masses (n,)) float64; reference signature :91."""
    positions = np.zeros((n, 3), dtype=np.float64)
    velocities = np.zeros((n, 3), dtype=np.float64)
    masses = np.ones(n, dtype=np.float64)

    if distribution == "galaxy":  # reference :104-146
        _disk_galaxy(positions, velocities, masses, R, G, n, R * 0.3, R * 0.03, R * 1.0, 0.012, 0.12, +1)
        com_vel = np.sum(velocities * masses[:, np.newaxis], axis=0) / np.sum(masses)
        velocities -= com_vel

    elif distribution == "collision":  # reference :148-232
        half = n // 2
        n2 = n - half
        scale_length = R * 0.25
        softening = R * 0.025
        separation = (R * 0.5) * 3.5
        _disk_galaxy(positions[:half], velocities[:half], masses[:half], R, G, half, scale_length, softening,
                     R * 0.5, 0.01, 0.10, +1, x0=-separation / 2)
        _disk_galaxy(positions[half:], velocities[half:], masses[half:], R, G, n2, scale_length, softening,
                     R * 0.5, 0.01, 0.10, -1, x0=separation / 2, y0=R * 0.15)
        total_mass = n * 0.001
        collision_speed = np.sqrt(2 * G * total_mass / separation) * 0.6
        velocities[:half, 0] += collision_speed
        velocities[half:, 0] -= collision_speed

    elif distribution == "cluster":  # Plummer sphere, reference :350-397
        a = R * 0.3
        u = np.random.uniform(0, 1, n)
        r = a / np.sqrt(u ** (-2 / 3) - 1)
        r = np.clip(r, 0, R * 1.5)
        phi = np.random.uniform(0, 2 * np.pi, n)
        cos_theta = np.random.uniform(-1, 1, n)
        sin_theta = np.sqrt(1 - cos_theta ** 2)
        positions[:, 0] = r * sin_theta * np.cos(phi)
        positions[:, 1] = r * cos_theta
        positions[:, 2] = r * sin_theta * np.sin(phi)
        total_mass = n * 0.001
        r_a_sq = (r / a) ** 2
        sigma_sq = G * total_mass / (6 * a) * (1 + r_a_sq) ** (-0.5)
        sigma = np.sqrt(np.maximum(sigma_sq, G * total_mass / (6 * a) * 0.01))
        # The reference draws (|normal|, uniform, uniform) per body in a Python loop; the scalar
        # draws interleave, so the stream must be consumed body by body to match it.
        scale = sigma * np.sqrt(3)
        for i in range(n):
            v_mag = np.abs(np.random.normal(0, scale[i]))
            v_phi = np.random.uniform(0, 2 * np.pi)
            v_cos = np.random.uniform(-1, 1)
            v_sin = np.sqrt(1 - v_cos ** 2)
            velocities[i, 0] = v_mag * v_sin * np.cos(v_phi)
            velocities[i, 1] = v_mag * v_cos
            velocities[i, 2] = v_mag * v_sin * np.sin(v_phi)
        com_vel = np.sum(velocities * masses[:, np.newaxis], axis=0) / np.sum(masses)
        velocities -= com_vel

    else:
        raise ValueError(f"distribution {distribution!r} is not part of this build "
                         f"(available: {sorted(DISTRIBUTIONS)})")
    return positions, velocities, masses


def generate_distribution_device(distribution: str, n: int, R: float, G: float, softening: float, damping: float = 1.0,
                                 theta: float = 0.5, seed: int = 42, device=None):
    """generate_distribution + backend construction in one step, on the GPU: returns a
    HIPBarnesHutSimulation whose bodies were drawn on the device (nbmi_create_generated; Philox
    stream, statistical parity with generate_distribution).  At 10 M bodies this replaces seconds of
    NumPy and a 640 MB upload by a few milliseconds."""
    from nbody.gpu_backend import HIPBarnesHutSimulation
    return HIPBarnesHutSimulation.generated(distribution, n, R, G, softening, damping, theta=theta, seed=seed,
                                            device=device)


def _preset(name, desc, cat, n, theta, G, eps, R, dist, frames, dtf, sub, fps, est):
    return {"name": name, "description": desc, "category": cat, "num_bodies": n, "theta": theta, "G": G,
            "softening": eps, "damping": 1.0, "spawn_radius": R, "distribution": dist, "total_frames": frames,
            "dt_per_frame": dtf, "substeps": sub, "target_fps": fps, "estimated_time": est}


# The presets BASELINE.json / SURVEY 8(d) name; constants as reference tools/presets.py
# :1774-1790, :1516-1532, :1552-1568, :1868-1884, :2424-2440.
PRESETS = {
    "quick_galaxy": _preset("Quick Galaxy", "Fast galaxy simulation for testing", "FAST", 100_000, 0.95, 0.15,
                            3.0, 500.0, "galaxy", 500, 0.2, 1, 30, "~25 seconds"),
    "4k_galaxy_1m": _preset("4K Galaxy 1M", "1 million body galaxy, ultra cinematic", "CINEMATIC_4K", 1_000_000,
                            0.5, 0.07, 1.5, 800.0, "galaxy", 3600, 0.05, 5, 60, "~11 hours"),
    "accurate_cluster": _preset("Globular Cluster", "Physically accurate globular cluster (Plummer model)",
                                "SCIENTIFIC", 200_000, 0.5, 0.05, 1.0, 300.0, "cluster", 2000, 0.08, 4, 24,
                                "~50 minutes"),
    "extreme_10m_collision": _preset("10 Million Collision", "Massive collision with 10M bodies", "EXTREME",
                                     10_000_000, 1.3, 0.08, 6.0, 2000.0, "collision", 500, 0.25, 1, 20,
                                     "~30 minutes"),
}


def get_preset_config(key: str) -> dict:
    """Preset dict + ``session_name`` (reference :2701-2709); None for unknown keys."""
    if key not in PRESETS:
        return None
    preset = PRESETS[key].copy()
    preset["session_name"] = key
    return preset
