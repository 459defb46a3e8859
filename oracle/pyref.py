"""ctypes front-end of the CPU oracle (oracle/nbref.c, oracle/bdref.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

The function names and flat-array signatures follow the reference's @njit functions
(nbody/simulation.py, boids/flock.py) so that tests read like calls into the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_TREE_NODES = 8_000_000  # nbody/simulation.py:35
UNCAPPED = 1 << 62

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i64 = C.c_int64
_dbl = C.c_double


def build(fast=False, native=False, out_dir=None):
    """Compile the oracle with gcc.  native=True adds -march=native (GPU-box local build)."""
    out_dir = out_dir or _HERE
    name = "libnbref_fast.so" if fast else "libnbref.so"
    out = os.path.join(out_dir, name)
    src = [os.path.join(_HERE, "nbref.c"), os.path.join(_HERE, "bdref.c")]
    if fast:
        flags = ["-O3", "-ffast-math", "-march=native" if native else "-march=x86-64-v3"]
    else:
        flags = ["-O2", "-ffp-contract=off", "-fno-fast-math"]
    cmd = ["gcc", *flags, "-fPIC", "-shared", "-fopenmp", "-o", out, *src, "-lm"]
    subprocess.run(cmd, check=True)
    return out


def _bind(lib):
    lib.nbref_num_threads.restype = C.c_int
    lib.nbref_set_num_threads.argtypes = [C.c_int]
    lib.nbref_compute_bounds.restype = _dbl
    lib.nbref_compute_bounds.argtypes = [_f64p, _i64]
    lib.nbref_build_octree.restype = _i64
    lib.nbref_build_octree.argtypes = [_f64p, _f64p, _i64, _dbl, _f64p, _f64p, _f64p, _f64p, _i32p, _i32p, _u8p,
                                       _i64, _i64]
    lib.nbref_compute_forces_bh.restype = None
    lib.nbref_compute_forces_bh.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i32p, _i32p, _u8p,
                                            _i64, _i64, _dbl, _dbl, _dbl, C.c_void_p]
    lib.nbref_update.restype = None
    lib.nbref_update.argtypes = [_f64p, _f64p, _f64p, _dbl, _dbl, _i64]
    lib.nbref_colors.restype = None
    lib.nbref_colors.argtypes = [_f64p, _f32p, _i64, _dbl]
    lib.nbref_direct_forces.restype = None
    lib.nbref_direct_forces.argtypes = [_f64p, _f64p, _f64p, _dbl, _dbl, _i64]
    lib.nbref_direct_forces_subset.restype = None
    lib.nbref_direct_forces_subset.argtypes = [_f64p, _f64p, _i64p, _i64, _f64p, _dbl, _dbl, _i64]
    lib.nbref_group_walk_hist.restype = None
    lib.nbref_group_walk_hist.argtypes = [_f64p, _i64p, _i64, C.c_int, _f64p, _f64p, _i32p, _u8p, _dbl, _dbl, _i64p,
                                          _i64p]
    lib.nbref_direct_update.restype = None
    lib.nbref_direct_update.argtypes = [_f64p, _f64p, _f64p, _dbl, _dbl, _i64]
    lib.nbref_step.restype = _i64
    lib.nbref_step.argtypes = [_f64p, _f64p, _f64p, _f64p, _i64, _dbl, _dbl, _dbl, _dbl, _dbl,
                               _f64p, _f64p, _f64p, _f64p, _i32p, _i32p, _u8p, _i64, _i64, C.c_void_p, C.c_void_p]
    lib.nbref_body_keys.restype = None
    lib.nbref_body_keys.argtypes = [_f64p, _i64, _dbl, _u64p, _u64p]
    lib.nbref_tree_cells.restype = C.c_int
    lib.nbref_tree_cells.argtypes = [_i32p, _i64, _i32p, _u64p]
    lib.nbref_group_walk_stats.restype = None
    lib.nbref_group_walk_stats.argtypes = [_f64p, _i64p, _i64, C.c_int, _f64p, _f64p, _i32p, _i32p, _u8p, _dbl,
                                           _dbl, _i64p]
    lib.nbref_visibility_points.restype = None
    lib.nbref_visibility_points.argtypes = [_f64p, _f64p, _dbl, _dbl, _dbl, _u8p, _i64]
    lib.bdref_visibility.restype = None
    lib.bdref_visibility.argtypes = [_f64p, _f64p, _dbl, _dbl, _dbl, _u8p, _i64]
    lib.bdref_build_vertices.restype = None
    lib.bdref_build_vertices.argtypes = [_f64p, _f64p, _f64p, _i32p, _f32p, _f32p, _dbl, _dbl, _i64]
    lib.bdref_assign_cells.restype = None
    lib.bdref_assign_cells.argtypes = [_f64p, _i32p, _dbl, C.c_int, _dbl, _i64]
    lib.bdref_build_cell_lists.restype = None
    lib.bdref_build_cell_lists.argtypes = [_i32p, _i32p, _i32p, _i32p, _i64, _i64]
    lib.bdref_sort_by_cell.restype = None
    lib.bdref_sort_by_cell.argtypes = [_i32p, _i32p, _i64, _i64]
    lib.bdref_compute_flocking_spatial.restype = None
    lib.bdref_compute_flocking_spatial.argtypes = [_f64p, _f64p, _f64p, _i32p, _i32p, _i32p, _f64p, _f64p, _f64p,
                                                   _f64p, _dbl, C.c_int, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl,
                                                   _dbl, _i64]
    lib.bdref_update_physics.restype = None
    lib.bdref_update_physics.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _dbl, _dbl, _dbl, _dbl,
                                         _dbl, _dbl, _i64]
    lib.bdref_step.restype = None
    lib.bdref_step.argtypes = [_f64p, _f64p, _f64p, _i64, _f64p, _dbl, C.c_void_p, _i32p, _i32p, _i32p, _i32p,
                               _f64p]
    return lib


_LIBS = {}


def lib(fast=False, path=None):
    key = path or ("fast" if fast else "strict")
    if key not in _LIBS:
        p = path or os.path.join(_HERE, "libnbref_fast.so" if fast else "libnbref.so")
        if not os.path.exists(p):
            build(fast=fast)
        _LIBS[key] = _bind(C.CDLL(p))
    return _LIBS[key]


class NodeArrays:
    """The seven flat octree arrays of the reference (nbody/simulation.py:478-484)."""

    def __init__(self, rows):
        self.rows = int(rows)
        self.centers = np.zeros((rows, 3))
        self.half = np.zeros(rows)
        self.mass = np.zeros(rows)
        self.com = np.zeros((rows, 3))
        self.children = np.full((rows, 8), -1, dtype=np.int32)
        self.body = np.full(rows, -1, dtype=np.int32)
        self.leaf = np.ones(rows, dtype=np.uint8)

    @classmethod
    def for_bodies(cls, n, rows=None):
        return cls(rows or min(MAX_TREE_NODES, max(4 * n, 64)))


def compute_bounds(pos, L=None):
    L = L or lib()
    return L.nbref_compute_bounds(np.ascontiguousarray(pos, np.float64), len(pos))


def build_octree(pos, masses, bounds, nd, cap=MAX_TREE_NODES, L=None):
    L = L or lib()
    nd.children.fill(-1)
    nd.body.fill(-1)
    nd.leaf.fill(1)
    nn = L.nbref_build_octree(pos, masses, len(pos), bounds, nd.centers, nd.half, nd.mass, nd.com, nd.children,
                              nd.body, nd.leaf, nd.rows, cap)
    if nn < 0:
        raise MemoryError("oracle node arrays too small")
    return int(nn)


def compute_forces_barnes_hut(pos, masses, nd, num_nodes, theta, G, softening, stats=False, L=None):
    L = L or lib()
    acc = np.zeros((len(pos), 3))
    st = np.zeros(6, dtype=np.int64)
    L.nbref_compute_forces_bh(pos, masses, acc, nd.centers, nd.half, nd.mass, nd.com, nd.children, nd.body,
                              nd.leaf, num_nodes, len(pos), theta, G, softening, st.ctypes.data)
    if stats:
        return acc, dict(visits=int(st[0]), accepted=int(st[1]), dropped=int(st[2]), peak_stack=int(st[3]),
                         opened=int(st[4]), chain_visits=int(st[5]))
    return acc


def update_positions_velocities(pos, vel, acc, damping, dt, L=None):
    (L or lib()).nbref_update(pos, vel, acc, damping, dt, len(pos))


def compute_colors_by_velocity(vel, max_speed, L=None):
    col = np.zeros((len(vel), 3), dtype=np.float32)
    (L or lib()).nbref_colors(np.ascontiguousarray(vel, np.float64), col, len(vel), max_speed)
    return col


def _cam(cam_pos, cam_forward, cam_right, cam_up):
    return np.ascontiguousarray(np.concatenate([cam_pos, cam_forward, cam_right, cam_up]), dtype=np.float64)


def compute_visibility_points(pos, cam_pos, cam_forward, cam_right, cam_up, tan_h, tan_v, far_dist, L=None):
    """nbody/simulation.py:403-434; returns the bool mask."""
    mask = np.zeros(len(pos), dtype=np.uint8)
    (L or lib()).nbref_visibility_points(np.ascontiguousarray(pos, np.float64),
                                         _cam(cam_pos, cam_forward, cam_right, cam_up), tan_h, tan_v, far_dist,
                                         mask, len(pos))
    return mask.astype(bool)


def compute_visibility_boids(pos, cam_pos, cam_forward, cam_right, cam_up, tan_h, tan_v, fog_end, L=None):
    """boids/flock.py:311-348 compute_visibility_numba; returns the bool mask."""
    mask = np.zeros(len(pos), dtype=np.uint8)
    (L or lib()).bdref_visibility(np.ascontiguousarray(pos, np.float64),
                                  _cam(cam_pos, cam_forward, cam_right, cam_up), tan_h, tan_v, fog_end, mask,
                                  len(pos))
    return mask.astype(bool)


def build_vertices(pos, vel, col, visible_indices, cone_length, cone_radius, L=None):
    """boids/flock.py:351-447 build_vertices_numba; returns (vertices, vert_colors) float32 (6k,3)."""
    k = len(visible_indices)
    verts = np.zeros((6 * k, 3), dtype=np.float32)
    vcols = np.zeros((6 * k, 3), dtype=np.float32)
    (L or lib()).bdref_build_vertices(np.ascontiguousarray(pos, np.float64), np.ascontiguousarray(vel, np.float64),
                                      np.ascontiguousarray(col, np.float64),
                                      np.ascontiguousarray(visible_indices, np.int32), verts, vcols,
                                      float(cone_length), float(cone_radius), k)
    return verts, vcols


def direct_forces(pos, masses, G, softening, L=None):
    acc = np.zeros((len(pos), 3))
    (L or lib()).nbref_direct_forces(pos, masses, acc, G, softening, len(pos))
    return acc


def direct_forces_subset(pos, masses, idx, G, softening, L=None):
    """compute_forces_brute_cuda's sum (nbody/gpu_backend.py:145-174) for the rows `idx` only."""
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    acc = np.zeros((len(idx), 3))
    (L or lib()).nbref_direct_forces_subset(pos, masses, idx, len(idx), acc, G, softening, len(pos))
    return acc


def direct_update(pos, vel, acc, dt, damping, L=None):
    (L or lib()).nbref_direct_update(pos, vel, acc, dt, damping, len(pos))


def body_keys(pos, bounds, L=None):
    hi = np.zeros(len(pos), dtype=np.uint64)
    lo = np.zeros(len(pos), dtype=np.uint64)
    (L or lib()).nbref_body_keys(np.ascontiguousarray(pos, np.float64), len(pos), bounds, hi, lo)
    return hi, lo


def tree_cells(nd, num_nodes, L=None):
    level = np.zeros(num_nodes, dtype=np.int32)
    key = np.zeros(num_nodes, dtype=np.uint64)
    rc = (L or lib()).nbref_tree_cells(nd.children, num_nodes, level, key)
    assert rc == 0, "tree not fully reachable from root"
    return level, key


class BHStepper:
    """Drives the CPU substep of tools/record.py:835-858 (bounds, fills, build, walk, update)."""

    def __init__(self, pos, vel, masses, theta, G, softening, damping, cap=MAX_TREE_NODES, rows=None, fast=False,
                 L=None):
        self.L = L or lib(fast=fast)
        self.pos = np.array(pos, dtype=np.float64, order="C")
        self.vel = np.array(vel, dtype=np.float64, order="C")
        self.masses = np.array(masses, dtype=np.float64, order="C")
        self.n = len(self.pos)
        self.acc = np.zeros((self.n, 3))
        self.theta, self.G, self.softening, self.damping = theta, G, softening, damping
        self.cap = cap
        self.nd = NodeArrays.for_bodies(self.n, rows)
        self.num_nodes = 0
        self.stats = np.zeros(6, dtype=np.int64)
        self.phase_s = np.zeros(5)

    def step(self, dt):
        nn = self.L.nbref_step(self.pos, self.vel, self.masses, self.acc, self.n, self.theta, self.G,
                               self.softening, self.damping, dt, self.nd.centers, self.nd.half, self.nd.mass,
                               self.nd.com, self.nd.children, self.nd.body, self.nd.leaf, self.nd.rows, self.cap,
                               self.stats.ctypes.data, self.phase_s.ctypes.data)
        if nn < 0:
            raise MemoryError("oracle node arrays too small")
        self.num_nodes = int(nn)
        return self.num_nodes


# ---- boids -----------------------------------------------------------------------------

BOIDS_DEFAULT = dict(bounds=500.0, wall_margin=3.0, wall_weight=10.0, max_speed=25.0, max_force=60.0,
                     perception_radius=5.0, separation_radius=3.0, separation_weight=2.5, alignment_weight=1.0,
                     cohesion_weight=1.0, color_blend_rate=1.0)  # config/boids.py:30-46
_BOID_KEYS = ["bounds", "wall_margin", "wall_weight", "max_speed", "max_force", "perception_radius",
              "separation_radius", "separation_weight", "alignment_weight", "cohesion_weight", "color_blend_rate"]


def boids_params(**over):
    d = dict(BOIDS_DEFAULT)
    d.update(over)
    return np.array([d[k] for k in _BOID_KEYS], dtype=np.float64)


def boids_grid(params):
    bounds, perception = params[0], params[5]
    cell = float(perception)
    dim = int(np.ceil(bounds * 2 / cell)) + 2
    return cell, dim, float(bounds + cell)


class FlockStepper:
    """Flock.update restated (boids/flock.py:627-678); use_numpy_argsort=True reproduces the
    reference's np.argsort permutation exactly (flock.py:618)."""

    def __init__(self, pos, vel, col, params, use_numpy_argsort=True, fast=False, L=None):
        self.L = L or lib(fast=fast)
        self.pos = np.array(pos, dtype=np.float64, order="C")
        self.vel = np.array(vel, dtype=np.float64, order="C")
        self.col = np.array(col, dtype=np.float64, order="C")
        self.params = np.array(params, dtype=np.float64)
        self.n = len(self.pos)
        self.cell, self.dim, self.offset = boids_grid(self.params)
        self.num_cells = self.dim ** 3
        self.cell_indices = np.zeros(self.n, dtype=np.int32)
        self.sorted_indices = np.arange(self.n, dtype=np.int32)
        self.cell_starts = np.zeros(self.num_cells, dtype=np.int32)
        self.cell_counts = np.zeros(self.num_cells, dtype=np.int32)
        self.f = np.zeros(12 * self.n)
        self.use_numpy_argsort = use_numpy_argsort

    @property
    def sep(self):
        return self.f[:3 * self.n].reshape(-1, 3)

    @property
    def ali(self):
        return self.f[3 * self.n:6 * self.n].reshape(-1, 3)

    @property
    def coh(self):
        return self.f[6 * self.n:9 * self.n].reshape(-1, 3)

    @property
    def avg(self):
        return self.f[9 * self.n:].reshape(-1, 3)

    def step(self, dt):
        sorted_in = None
        if self.use_numpy_argsort:
            self.L.bdref_assign_cells(self.pos, self.cell_indices, self.cell, self.dim, self.offset, self.n)
            order = np.ascontiguousarray(np.argsort(self.cell_indices).astype(np.int32))
            sorted_in = order.ctypes.data
            self._keep = order
        self.L.bdref_step(self.pos, self.vel, self.col, self.n, self.params, float(dt), sorted_in,
                          self.cell_indices, self.sorted_indices, self.cell_starts, self.cell_counts, self.f)
