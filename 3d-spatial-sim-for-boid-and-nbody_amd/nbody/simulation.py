"""NBodySimulation - the reference's simulation object (nbody/simulation.py:441-964) on the HIP
backend.  Same constructor, public attributes and ``update(dt)``; all physics runs on the GPU
through nbody.gpu_backend (there is no CPU Barnes-Hut in this package, so a missing HIP device
is an error, not a fallback).  ``draw`` (OpenGL, reference :905) is out of scope.
"""
import math

import numpy as np

from config import nbody as config

from .gpu_backend import Backend, create_gpu_simulation, get_backend

MAX_TREE_NODES = 8_000_000  # reference :35 (the HIP build allocates 4N rows like the reference :477)


def _disk(n, scale, R, rng=np.random):
    """r ~ Exp(scale), theta ~ U(0, 2pi): the first two draws of every live disk generator."""
    r = rng.exponential(scale, n)
    theta = rng.uniform(0, 2 * np.pi, n)
    return r, theta


def _ic_galaxy(n, R, G):
    """Rotating exponential disk, thin in y, 0.1% heavy bodies (reference :557-588)."""
    r, theta = _disk(n, R * 0.3, R)
    z = np.random.normal(0, R * 0.02, n)
    pos = np.zeros((n, 3))
    pos[:, 0] = r * np.cos(theta)
    pos[:, 1] = z
    pos[:, 2] = r * np.sin(theta)
    vel = np.zeros((n, 3))
    speed = np.sqrt(G * n * 0.001 / (r + 1.0))
    vel[:, 0] = -speed * np.sin(theta)
    vel[:, 2] = speed * np.cos(theta)
    vel[:, 1] = np.random.normal(0, speed * 0.1, n)
    m = np.ones(n)
    m[np.random.choice(n, max(1, n // 1000), replace=False)] = 100.0
    return pos, vel, m


def _ic_spiral(n, R, G):
    """Central mass + flattened bulge + 4-arm logarithmic spiral disk (reference :590-677)."""
    pos = np.zeros((n, 3))
    vel = np.zeros((n, 3))
    m = np.ones(n)
    central = n * 50.0
    m[0] = central
    nb = max(1, n // 20)
    br = np.random.exponential(R * 0.05, nb)
    bt = np.random.uniform(0, 2 * np.pi, nb)
    bp = np.arccos(np.random.uniform(-1, 1, nb))
    pos[1:nb + 1, 0] = br * np.sin(bp) * np.cos(bt)
    pos[1:nb + 1, 1] = br * np.sin(bp) * np.sin(bt) * 0.3
    pos[1:nb + 1, 2] = br * np.cos(bp)
    bo = np.sqrt(G * central / (br + 1.0)) * 0.5
    vel[1:nb + 1, 0] = np.random.normal(0, bo * 0.3, nb)
    vel[1:nb + 1, 1] = np.random.normal(0, bo * 0.1, nb)
    vel[1:nb + 1, 2] = np.random.normal(0, bo * 0.3, nb)
    d0 = nb + 1
    nd = n - d0
    dr = np.clip(np.random.exponential(R * 0.25, nd), R * 0.02, R * 0.9)
    base = np.log(dr / (R * 0.05) + 1) / 0.3
    arm = np.random.randint(0, 4, nd) * (2 * np.pi / 4)
    scatter = np.random.normal(0, 0.3, nd)
    dth = base + arm + scatter
    dz = np.random.normal(0, R * 0.01, nd) * (1 + dr / R)
    pos[d0:, 0] = dr * np.cos(dth)
    pos[d0:, 1] = dz
    pos[d0:, 2] = dr * np.sin(dth)
    enclosed = central + dr / R * n * 0.5
    speed = np.sqrt(G * enclosed / (dr + 0.1))
    vel[d0:, 0] = -speed * np.sin(dth)
    vel[d0:, 2] = speed * np.cos(dth)
    vel[d0:, 0] += np.random.normal(0, speed * 0.05, nd)
    vel[d0:, 1] = np.random.normal(0, speed * 0.02, nd)
    vel[d0:, 2] += np.random.normal(0, speed * 0.05, nd)
    return pos, vel, m


def _ic_sphere(n, R, G):
    """Uniform ball of radius 0.8 R, small random velocities (reference :679-700)."""
    phi = np.random.uniform(0, 2 * np.pi, n)
    ct = np.random.uniform(-1, 1, n)
    st = np.sqrt(1 - ct ** 2)
    r = R * 0.8 * np.cbrt(np.random.uniform(0, 1, n))
    pos = np.zeros((n, 3))
    pos[:, 0] = r * st * np.cos(phi)
    pos[:, 1] = r * st * np.sin(phi)
    pos[:, 2] = r * ct
    vel = np.random.normal(0, 0.5, (n, 3)).astype(np.float64)
    return pos, vel, np.ones(n)


def _ic_collision(n, R, G):
    """Two disks at x = -/+0.4 R approaching at +/-2 (reference :702-735)."""
    half = n // 2
    pos = np.zeros((n, 3))
    vel = np.zeros((n, 3))
    for sl, cnt, x0, vx0 in ((slice(0, half), half, -R * 0.4, 2.0), (slice(half, n), n - half, R * 0.4, -2.0)):
        r = np.random.exponential(R * 0.2, cnt)
        th = np.random.uniform(0, 2 * np.pi, cnt)
        pos[sl, 0] = r * np.cos(th) + x0
        pos[sl, 1] = np.random.normal(0, R * 0.02, cnt)
        pos[sl, 2] = r * np.sin(th)
        speed = np.sqrt(G * cnt * 0.001 / (r + 1.0))
        vel[sl, 0] = -speed * np.sin(th) + vx0
        vel[sl, 2] = speed * np.cos(th)
    return pos, vel, np.ones(n)


def _ic_uniform(n, R, G):
    """Uniform cube of half width 0.8 R (reference :737-746)."""
    pos = ((np.random.rand(n, 3) - 0.5) * 2 * R * 0.8).astype(np.float64)
    vel = np.random.normal(0, 1.0, (n, 3)).astype(np.float64)
    return pos, vel, np.ones(n)


_LIVE_ICS = {"galaxy": _ic_galaxy, "spiral": _ic_spiral, "sphere": _ic_sphere, "collision": _ic_collision}


class NBodySimulation:
    """Drop-in for reference nbody.NBodySimulation (:441): ``NBodySimulation(num_bodies)``,
    ``update(dt)``, attrs positions/velocities/masses/accelerations/colors/num_bodies/
    _num_tree_nodes/_visible_count/theta/G/softening/damping/current_bounds."""

    def __init__(self, num_bodies: int = 1_000_000, seed=None):
        self.num_bodies = num_bodies
        cfg = config.NBODY
        self.spawn_radius = float(cfg["spawn_radius"])
        self.G = float(cfg["G"])
        self.theta = float(cfg["theta"])
        self.softening = float(cfg["softening"])
        self.damping = float(cfg["damping"])
        self.point_size = float(cfg["point_size"])
        self.max_speed_color = float(cfg["max_speed_color"])
        self._tree_facts = {"num_nodes": 0, "bounds": self.spawn_radius * 2}
        self._tree_facts_stale = False
        if seed is not None:  # the reference never seeds; extra kwarg for reproducible runs
            np.random.seed(seed)
        gen = _LIVE_ICS.get(cfg.get("distribution", "galaxy"), _ic_uniform)
        self.positions, self.velocities, self.masses = gen(num_bodies, self.spawn_radius, self.G)
        self.accelerations = np.zeros((num_bodies, 3), dtype=np.float64)
        self.colors = np.zeros((num_bodies, 3), dtype=np.float32)
        self._gpu_sim = None
        self._use_gpu = False
        self._backend = None
        self._init_gpu_backend()
        self._visible_mask = np.ones(num_bodies, dtype=np.bool_)
        self._visible_count = num_bodies
        self.fog_end = float(config.CAMERA["far_clip"])
        print(f"[NBody] Initialized {num_bodies:,} bodies")

    def _init_gpu_backend(self):
        backend, info = get_backend()
        if backend != Backend.HIP:
            raise RuntimeError(f"[NBody] no HIP backend ({info}); this build has no CPU fallback")
        self._gpu_sim = create_gpu_simulation(self.positions, self.velocities, self.masses, self.G, self.softening,
                                              self.damping, theta=self.theta)
        if self._gpu_sim is None:
            raise RuntimeError("[NBody] create_gpu_simulation returned None; this build has no CPU fallback")
        self._use_gpu = True
        self._backend = backend
        print(f"[NBody] GPU acceleration enabled: {backend.value}")

    def update(self, dt: float):
        """One timestep; dt capped at 0.02 exactly as the reference (:799-807)."""
        dt = min(dt, 0.02)
        self._update_gpu(dt)

    def _update_gpu(self, dt: float):
        # reference :809-817: step, colours, then D2H of positions (as f64) and colours
        self._gpu_sim.step(dt)
        self._gpu_sim.compute_colors(self.max_speed_color)
        self.positions = self._gpu_sim.get_positions().astype(np.float64)
        self.colors = self._gpu_sim.get_colors()
        self._tree_facts_stale = True  # fetched when somebody reads them (HUD), not once per frame

    def _tree_fact(self, key):
        if self._tree_facts_stale and hasattr(self._gpu_sim, "tree_stats"):
            self._tree_facts = self._gpu_sim.tree_stats(depth=False)
            self._tree_facts_stale = False
        return self._tree_facts[key]

    @property
    def _num_tree_nodes(self):
        """HUD number of the reference (nbody_main.py:153): nodes of the last step's octree."""
        return self._tree_fact("num_nodes")

    @property
    def current_bounds(self):
        """Root half size of the last step's octree (reference attribute, simulation.py:822)."""
        return self._tree_fact("bounds")

    def sync_velocities(self):
        """Fetch velocities from the device (the reference's GPU path never refreshes them)."""
        self.velocities = self._gpu_sim.get_velocities()
        return self.velocities

    def _compute_visibility(self, cam_pos, cam_forward, cam_right, cam_up, fov_v, aspect):
        """Reference :880-903, on the device: sets _visible_count and keeps the compacted float32
        positions / colours a viewer would upload (`visible_pos`, `visible_colors`)."""
        half_fov_v = fov_v / 2
        half_fov_h = math.atan(math.tan(half_fov_v) * aspect)
        self.visible_pos, self.visible_colors = self._gpu_sim.visible_points(
            cam_pos, cam_forward, cam_right, cam_up, math.tan(half_fov_h), math.tan(half_fov_v), self.fog_end)
        self._visible_count = len(self.visible_pos)

    def visible_arrays(self, cam_pos=None, cam_forward=None, cam_right=None, cam_up=None, fov=None, aspect=None):
        """The two arrays draw() (reference :905-928) hands to its VBOs: (positions[mask] float32,
        colors[mask]); everything is visible without a camera."""
        if cam_pos is None:
            self._visible_count = self.num_bodies
            return self.positions.astype(np.float32), self.colors
        fov_rad = math.radians(fov) if fov else math.radians(75)
        self._compute_visibility(cam_pos, cam_forward, cam_right, cam_up, fov_rad, aspect if aspect else (16 / 9))
        return self.visible_pos, self.visible_colors

    def draw(self, *args, **kwargs):
        raise NotImplementedError("OpenGL rendering (reference nbody/simulation.py:905) is out of scope of this "
                                  "build; visible_arrays() returns what draw() would upload")
