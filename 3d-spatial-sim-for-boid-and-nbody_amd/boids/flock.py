"""Flock - the reference's boids object (boids/flock.py:454-782) on the HIP backend.

Same constructor and ``update(dt)``; ``positions`` / ``velocities`` / ``colors`` are (N,3)
float64 arrays like the reference's, fetched from the device lazily (state lives in HBM
between updates).  One ``update`` = assign_cells -> sort -> cell table -> 27-cell sweep ->
physics (reference :627-678) in bdmi_step.  ``draw`` (OpenGL, :730) is out of scope.
"""
import ctypes as C

import math

import numpy as np

import nbmi_native as _nat
from config import boids as config


def rainbow_colors(count: int) -> np.ndarray:
    """Shuffled HSV rainbow, s=0.9 v=1.0, float32 maths (reference _generate_colors :587-608).
    Consumes the global NumPy RNG exactly once (np.random.shuffle)."""
    hues = np.linspace(0, 1, count, endpoint=False, dtype=np.float32)
    np.random.shuffle(hues)
    s, v = 0.9, 1.0
    h6 = hues * 6.0
    sector = h6.astype(np.int32) % 6
    f = h6 - np.floor(h6)
    p = v * (1.0 - s)
    q = v * (1.0 - s * f)
    t = v * (1.0 - s * (1.0 - f))
    vv = np.full_like(q, v)
    pp = np.full_like(q, p)
    table = ((vv, t, pp), (q, vv, pp), (pp, vv, t), (pp, q, vv), (t, pp, vv), (vv, pp, q))
    out = np.zeros((count, 3), dtype=np.float32)
    for k, chans in enumerate(table):
        mask = sector == k
        for c in range(3):
            out[mask, c] = chans[c][mask]
    return out


def generate_initial_state(num_boids: int, bounds: float, max_speed: float):
    """positions U(-b,b)^3, velocities U(-max_speed/2, max_speed/2)^3, rainbow colours - global
    NumPy RNG, call order of reference Flock.__init__ :488-490."""
    pos = ((np.random.rand(num_boids, 3) - 0.5) * 2 * bounds).astype(np.float64)
    vel = ((np.random.rand(num_boids, 3) - 0.5) * max_speed).astype(np.float64)
    col = rainbow_colors(num_boids).astype(np.float64)
    return pos, vel, col


class Flock:
    """Drop-in for reference boids.Flock (:454): ``Flock(num_boids)``, ``update(dt)``."""

    def __init__(self, num_boids: int = 1000, seed=None, device: int = 0):
        self.num_boids = num_boids
        B = config.BOIDS
        self.bounds = np.float64(B["bounds"])
        self.wall_margin = np.float64(B["wall_margin"])
        self.wall_weight = np.float64(B["wall_weight"])
        self.max_speed = np.float64(B["max_speed"])
        self.max_force = np.float64(B["max_force"])
        self.perception_radius = np.float64(B["perception_radius"])
        self.separation_radius = np.float64(B["separation_radius"])
        self.separation_weight = np.float64(B["separation_weight"])
        self.alignment_weight = np.float64(B["alignment_weight"])
        self.cohesion_weight = np.float64(B["cohesion_weight"])
        self.color_blend_rate = np.float64(B["color_blend_rate"])
        # spatial grid exactly as reference :478-481
        self.cell_size = float(self.perception_radius)
        self.grid_dim = int(np.ceil(self.bounds * 2 / self.cell_size)) + 2
        self.num_cells = self.grid_dim ** 3
        self.grid_offset = float(self.bounds + self.cell_size)
        self.fog_end = float(config.CAMERA["far_clip"])
        self.fov_margin = 1.15
        self.cone_length = np.float32(B["size"])          # reference :466-467
        self.cone_radius = np.float32(B["size"] * 0.35)
        self.verts_per_boid = 6
        if seed is not None:  # the reference is unseeded; extra kwarg
            np.random.seed(seed)
        pos, vel, col = generate_initial_state(num_boids, self.bounds, self.max_speed)
        self._visible_count = num_boids
        self._lib = _nat.load()
        params = np.array([float(B[k]) for k in config.PARAM_ORDER], dtype=np.float64)
        self._h = self._lib.bdmi_create(num_boids, _nat.ptr(pos), _nat.ptr(vel), _nat.ptr(col), _nat.ptr(params),
                                        int(device))
        if not self._h:
            raise RuntimeError(f"bdmi_create failed: {_nat.last_error()} (this build has no CPU fallback)")
        self._cache = {"positions": pos, "velocities": vel, "colors": col}

    # ---- state access (reference attributes) ----------------------------------------------
    def _fetch(self):
        if self._cache is None:
            n = self.num_boids
            pos, vel, col = (np.empty((n, 3)) for _ in range(3))
            _nat.check(self._lib.bdmi_get_state(self._h, _nat.ptr(pos), _nat.ptr(vel), _nat.ptr(col)),
                       "bdmi_get_state")
            self._cache = {"positions": pos, "velocities": vel, "colors": col}
        return self._cache

    @property
    def positions(self):
        return self._fetch()["positions"]

    @property
    def velocities(self):
        return self._fetch()["velocities"]

    @property
    def colors(self):
        return self._fetch()["colors"]

    def set_state(self, positions=None, velocities=None, colors=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in
                (positions, velocities, colors)]
        for a in arrs:
            if a is not None and a.shape != (self.num_boids, 3):
                raise ValueError("state arrays must be (N,3)")
        _nat.check(self._lib.bdmi_set_state(self._h, *[_nat.ptr(a) for a in arrs]), "bdmi_set_state")
        self._cache = None

    # ---- the hot path -----------------------------------------------------------------------
    def update(self, dt: float, substeps: int = 1):
        """Flock.update(dt) (reference :627); no dt cap here - callers cap at 0.05."""
        _nat.check(self._lib.bdmi_step(self._h, float(np.float64(dt)), int(substeps)), "bdmi_step")
        self._cache = None

    def sync(self):
        _nat.check(self._lib.bdmi_sync(self._h), "bdmi_sync")

    # ---- parity hooks -------------------------------------------------------------------------
    def cell_indices(self) -> np.ndarray:
        out = np.empty(self.num_boids, dtype=np.int32)
        _nat.check(self._lib.bdmi_get_cell_indices(self._h, _nat.ptr(out)), "bdmi_get_cell_indices")
        return out

    def forces(self):
        """(separation, alignment, cohesion, avg_colors) for the current state, no integration."""
        n = self.num_boids
        outs = [np.empty((n, 3)) for _ in range(4)]
        _nat.check(self._lib.bdmi_get_forces(self._h, *[_nat.ptr(a) for a in outs]), "bdmi_get_forces")
        return tuple(outs)

    def grid_info(self):
        dim, cells, occ = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        _nat.check(self._lib.bdmi_grid_info(self._h, C.addressof(dim), C.addressof(cells), C.addressof(occ)),
                   "bdmi_grid_info")
        return dict(grid_dim=int(dim.value), num_cells=int(cells.value), occupied=int(occ.value))

    def enable_timers(self, on=True):
        _nat.check(self._lib.bdmi_enable_timers(self._h, 1 if on else 0), "bdmi_enable_timers")

    def timers(self, reset=False):
        ms = np.zeros(3)
        cnt = C.c_int64(0)
        _nat.check(self._lib.bdmi_get_timers(self._h, _nat.ptr(ms), C.addressof(cnt), 1 if reset else 0),
                   "bdmi_get_timers")
        return dict(sort_ms=ms[0], table_ms=ms[1], sweep_ms=ms[2], steps=int(cnt.value))

    # ---- render-side reduction (reference :680-728) on the device -----------------------------
    def visible_vertices(self, cam_pos=None, cam_forward=None, cam_right=None, cam_up=None, fov=None, aspect=None):
        """What draw() (reference :730-756) hands to its VBOs: (vertices, vert_colors) float32,
        6 rows per visible boid in ascending boid order; sets _visible_count.  Frustum test
        (compute_visibility_numba), compaction and cone building (build_vertices_numba) run on
        the GPU, only the visible part is copied back."""
        n = self.num_boids
        if cam_pos is None:  # reference: everything visible
            cam = np.array([0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0], dtype=np.float64)
            cam[0:3] = (0.0, 0.0, -1e300)
            tan_h = tan_v = float("inf")
            fog = float("inf")
        else:
            fov_rad = math.radians(fov) if fov else math.radians(75)
            aspect = aspect if aspect else (16 / 9)
            half_fov_v = (fov_rad / 2) * self.fov_margin
            half_fov_h = math.atan(math.tan(half_fov_v) * aspect)
            tan_h, tan_v, fog = math.tan(half_fov_h), math.tan(half_fov_v), self.fog_end
            cam = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(3) for a in
                                                       (cam_pos, cam_forward, cam_right, cam_up)]))
        if getattr(self, "_vertices", None) is None:
            self._vertices = np.zeros((n * 6, 3), dtype=np.float32)
            self._vert_colors = np.zeros((n * 6, 3), dtype=np.float32)
        cnt = C.c_int64(0)
        _nat.check(self._lib.bdmi_visible_vertices(self._h, _nat.ptr(cam), tan_h, tan_v, fog, float(self.cone_length),
                                                   float(self.cone_radius), _nat.ptr(self._vertices),
                                                   _nat.ptr(self._vert_colors), n, C.addressof(cnt)),
                   "bdmi_visible_vertices")
        self._visible_count = int(cnt.value)
        k = self._visible_count * self.verts_per_boid
        return self._vertices[:k], self._vert_colors[:k]

    def draw(self, *args, **kwargs):
        raise NotImplementedError("OpenGL rendering (reference boids/flock.py:730) is out of scope of this build; "
                                  "visible_vertices() returns what draw() would upload")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bdmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
