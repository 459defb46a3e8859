"""Backend seam of the N-body simulation - MI355X / HIP edition.

Mirrors the reference's nbody/gpu_backend.py: the ``Backend`` enum (:29-33) gains a ``HIP``
member that ``detect_backend`` (:36-55) tries first; ``get_backend`` / ``force_backend``
(:119-132) keep their cached-global behaviour; ``create_gpu_simulation`` (:623-679) returns an
object speaking the reference's backend protocol

    step(dt) / compute_colors(max_speed) / get_positions() -> (N,3) f32 /
    get_velocities() -> (N,3) f64 / get_colors() -> (N,3) f32 / sync()

(reference class CUDASimulation, :336-409) or ``None`` when no HIP device exists - the
reference's own "fall back" convention.  This package has no CPU path to fall back to, so its
callers (NBodySimulation, record) raise on ``None``.
"""
import ctypes as C
import os
from enum import Enum
from typing import Optional, Tuple

import numpy as np

import nbmi_native as _nat

METHOD_BARNES_HUT = 0
METHOD_DIRECT = 1


class Backend(Enum):
    HIP = "hip"                    # MI355X: Barnes-Hut (default) or direct N^2 in hand-written HIP
    CUDA = "cuda"                  # reference members kept so `Backend.X` comparisons still work
    METAL_BH = "metal_barnes_hut"
    METAL = "metal"
    CPU = "cpu"


def _check_hip() -> Tuple[bool, str]:
    try:
        n = _nat.device_count()
    except (ImportError, OSError, AttributeError) as e:  # library missing / stale
        return False, f"libnbmi.so unavailable: {e}"
    if n > 0:
        return True, f"{n} HIP device(s), gfx950 kernels (libnbmi.so)"
    return False, "no HIP device"


def detect_backend() -> Tuple[Backend, str]:
    ok, info = _check_hip()
    if ok:
        return Backend.HIP, info
    return Backend.CPU, info


_BACKEND: Optional[Backend] = None
_BACKEND_INFO: str = ""


def get_backend() -> Tuple[Backend, str]:
    """Current backend, detected once and cached (reference :119-125)."""
    global _BACKEND, _BACKEND_INFO
    if _BACKEND is None:
        _BACKEND, _BACKEND_INFO = detect_backend()
        print(f"[GPU] Using backend: {_BACKEND.value} - {_BACKEND_INFO}")
    return _BACKEND, _BACKEND_INFO


def force_backend(backend: Backend):
    """Pin the backend (reference :128-132)."""
    global _BACKEND, _BACKEND_INFO
    _BACKEND = backend
    _BACKEND_INFO = f"Forced: {backend.value}"


def _as_f64(a, shape_tail):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape[1:] != shape_tail:
        raise ValueError(f"expected array of shape (N,{','.join(map(str, shape_tail))}), got {a.shape}")
    return a


class _HIPSimulation:
    """Shared implementation of the backend protocol on top of the nbmi_* C ABI."""

    _method = METHOD_BARNES_HUT

    def __init__(self, positions, velocities, masses, G, softening, damping, theta=0.5, device=None):
        lib = _nat.load()
        pos = _as_f64(positions, (3,))
        vel = _as_f64(velocities, (3,))
        m = _as_f64(masses, ())
        if not (len(pos) == len(vel) == len(m)):
            raise ValueError("positions, velocities and masses must have the same length")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, _nat.device_count())
        self.n = len(pos)
        self.G, self.softening, self.damping, self.theta = float(G), float(softening), float(damping), float(theta)
        self.device = int(device)
        self._lib = lib
        self._h = lib.nbmi_create(self.n, _nat.ptr(pos), _nat.ptr(vel), _nat.ptr(m), self.G, self.softening,
                                  self.damping, self.theta, self._method, self.device)
        if not self._h:
            raise RuntimeError(f"nbmi_create failed: {_nat.last_error()}")
        kind = "Barnes-Hut" if self._method == METHOD_BARNES_HUT else "direct N^2"
        print(f"[HIP] Initialized with {self.n:,} bodies ({kind}) on device {self.device}")

    @classmethod
    def generated(cls, distribution, n, spawn_radius, G, softening, damping, theta=0.5, seed=42, device=None):
        """Same backend object, but the bodies are drawn ON THE DEVICE from the reference's
        generate_distribution formulas (tools/presets.py:104-232, :350-397; `distribution` in
        "galaxy" / "collision" / "cluster") with a Philox stream keyed by `seed`: statistical,
        not bit, parity with the NumPy generator; no host arrays, no upload."""
        kinds = {"galaxy": 0, "collision": 1, "cluster": 2}
        if distribution not in kinds:
            raise ValueError(f"device-side generator has {sorted(kinds)}, not {distribution!r}")
        self = cls.__new__(cls)
        lib = _nat.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, _nat.device_count())
        self.n = int(n)
        self.G, self.softening, self.damping, self.theta = float(G), float(softening), float(damping), float(theta)
        self.device = int(device)
        self._lib = lib
        self._h = lib.nbmi_create_generated(kinds[distribution], self.n, float(spawn_radius), int(seed) & (2 ** 64 - 1),
                                            self.G, self.softening, self.damping, self.theta, cls._method, self.device)
        if not self._h:
            raise RuntimeError(f"nbmi_create_generated failed: {_nat.last_error()}")
        print(f"[HIP] Generated {self.n:,} bodies ({distribution}, seed {seed}) on device {self.device}")
        return self

    def get_masses(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        _nat.check(self._lib.nbmi_get_masses_f64(self._h, _nat.ptr(out)), "nbmi_get_masses_f64")
        return out

    # ---- reference protocol -------------------------------------------------------------
    def step(self, dt: float):
        _nat.check(self._lib.nbmi_step(self._h, float(dt), 1), "nbmi_step")

    def compute_colors(self, max_speed: float):
        _nat.check(self._lib.nbmi_compute_colors(self._h, float(max_speed)), "nbmi_compute_colors")

    def get_positions(self) -> np.ndarray:
        out = np.empty((self.n, 3), dtype=np.float32)
        _nat.check(self._lib.nbmi_get_positions_f32(self._h, _nat.ptr(out)), "nbmi_get_positions_f32")
        return out

    def get_velocities(self) -> np.ndarray:
        out = np.empty((self.n, 3), dtype=np.float64)
        _nat.check(self._lib.nbmi_get_velocities_f64(self._h, _nat.ptr(out)), "nbmi_get_velocities_f64")
        return out

    def get_colors(self) -> np.ndarray:
        out = np.empty((self.n, 3), dtype=np.float32)
        _nat.check(self._lib.nbmi_get_colors_f32(self._h, _nat.ptr(out)), "nbmi_get_colors_f32")
        return out

    def sync(self):
        _nat.check(self._lib.nbmi_sync(self._h), "nbmi_sync")

    # ---- supersets ----------------------------------------------------------------------
    def step_many(self, dt: float, substeps: int):
        """`substeps` steps enqueued back to back without host round trips."""
        _nat.check(self._lib.nbmi_step(self._h, float(dt), int(substeps)), "nbmi_step")

    def step_count(self) -> int:
        """Steps the device has been asked to take since the handle was created (counted by the library)."""
        return int(self._lib.nbmi_step_count(self._h))

    def get_positions_f64(self) -> np.ndarray:
        out = np.empty((self.n, 3), dtype=np.float64)
        _nat.check(self._lib.nbmi_get_positions_f64(self._h, _nat.ptr(out)), "nbmi_get_positions_f64")
        return out

    def set_state(self, positions, velocities):
        pos = _as_f64(positions, (3,))
        vel = _as_f64(velocities, (3,))
        if len(pos) != self.n or len(vel) != self.n:
            raise ValueError("state arrays must have N rows")
        _nat.check(self._lib.nbmi_set_state(self._h, _nat.ptr(pos), _nat.ptr(vel)), "nbmi_set_state")

    def accelerations(self) -> np.ndarray:
        """Accelerations of the current positions (force pass only, no integration)."""
        out = np.empty((self.n, 3), dtype=np.float64)
        _nat.check(self._lib.nbmi_get_accelerations_f64(self._h, _nat.ptr(out)), "nbmi_get_accelerations_f64")
        return out

    def visible_points(self, cam_pos, cam_forward, cam_right, cam_up, tan_h, tan_v, far_dist):
        """Frustum culling + compaction on the device (reference compute_visibility_points,
        nbody/simulation.py:403-434, and the gather of draw(), :927-928): returns
        (positions[mask] float32, colors[mask] float32) in body order - only these rows leave the GPU.
        The two arrays are views into buffers this object re-uses: valid until the next call."""
        cam = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(3) for a in
                                                   (cam_pos, cam_forward, cam_right, cam_up)]))
        cnt = C.c_int64(0)
        if not hasattr(self, "_vis_buf") or self._vis_buf[0].shape[0] != self.n:
            self._vis_buf = (np.empty((self.n, 3), dtype=np.float32), np.empty((self.n, 3), dtype=np.float32))
        p, c = self._vis_buf
        _nat.check(self._lib.nbmi_visible_points(self._h, _nat.ptr(cam), float(tan_h), float(tan_v), float(far_dist),
                                                 _nat.ptr(p), _nat.ptr(c), self.n, C.addressof(cnt)),
                   "nbmi_visible_points")
        k = int(cnt.value)
        return p[:k], c[:k]

    # frame codec on the device (tools/record.py: format-1 / format-2 payloads of a .zstd frame)
    def frame_keyframe(self):
        """(positions f32 (N,3), colours f32 (N,3)); they become the device's previous decoded frame."""
        p, c = np.empty((self.n, 3), dtype=np.float32), np.empty((self.n, 3), dtype=np.float32)
        _nat.check(self._lib.nbmi_frame_keyframe(self._h, _nat.ptr(p), _nat.ptr(c)), "nbmi_frame_keyframe")
        return p, c

    def frame_delta(self):
        """(int16 (N,3) position deltas, int16 (N,3) colour deltas) = int16((cur - prev) * 1000) against the
        previous DECODED frame kept on the device (reference tools/record.py:254-262), 12 B/body over PCIe."""
        dp, dc = np.empty((self.n, 3), dtype=np.int16), np.empty((self.n, 3), dtype=np.int16)
        _nat.check(self._lib.nbmi_frame_delta_i16(self._h, _nat.ptr(dp), _nat.ptr(dc)), "nbmi_frame_delta_i16")
        return dp, dc

    def frame_set_previous(self, positions, colors):
        p = np.ascontiguousarray(positions, dtype=np.float32)
        c = np.ascontiguousarray(colors, dtype=np.float32)
        _nat.check(self._lib.nbmi_frame_set_previous(self._h, _nat.ptr(p), _nat.ptr(c)), "nbmi_frame_set_previous")

    # multi-GPU row exchange (device pointers; see nbody/sharded.py)
    def set_shard(self, begin, end):
        _nat.check(self._lib.nbmi_set_shard(self._h, int(begin), int(end)), "nbmi_set_shard")

    def set_exchange_sync(self, sync: bool):
        _nat.check(self._lib.nbmi_set_exchange_sync(self._h, 1 if sync else 0), "nbmi_set_exchange_sync")

    def export_shard(self, dev_ptr):
        _nat.check(self._lib.nbmi_export_shard(self._h, int(dev_ptr)), "nbmi_export_shard")

    def import_ranks(self, dev_ptr, begin, end):
        _nat.check(self._lib.nbmi_import_ranks(self._h, int(dev_ptr), int(begin), int(end)), "nbmi_import_ranks")

    def enable_timers(self, on=True):
        _nat.check(self._lib.nbmi_enable_timers(self._h, 1 if on else 0), "nbmi_enable_timers")

    def timers(self, reset=False):
        ms = np.zeros(5)
        cnt = C.c_int64(0)
        _nat.check(self._lib.nbmi_get_timers(self._h, _nat.ptr(ms), C.addressof(cnt), 1 if reset else 0),
                   "nbmi_get_timers")
        names = ("keys_ms", "sort_ms", "tree_ms", "walk_ms", "other_ms")
        d = dict(zip(names, ms.tolist()))
        d["steps"] = int(cnt.value)
        return d

    def stream_handle(self) -> int:
        return int(self._lib.nbmi_stream(self._h) or 0)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nbmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HIPBarnesHutSimulation(_HIPSimulation):
    """Octree build + tree walk + kick-drift on the GPU (reference CPU path
    nbody/simulation.py:63-305 behind the protocol of MetalBarnesHutSimulation,
    nbody/metal/metal_backend.py:246)."""

    _method = METHOD_BARNES_HUT

    def build_tree(self):
        _nat.check(self._lib.nbmi_build_tree(self._h), "nbmi_build_tree")

    def force_precision_share(self):
        """(share of the waves whose own density asked for float64 forces in the last step, every wave float64?)"""
        share, all64 = C.c_double(0.0), C.c_int(0)
        _nat.check(self._lib.nbmi_force_precision_share(self._h, C.addressof(share), C.addressof(all64)),
                   "nbmi_force_precision_share")
        return float(share.value), bool(all64.value)

    FORCE_PRECISION = {"auto": 0, "f32": 1, "f64": 2}

    def set_force_precision(self, mode="auto", tau=0.0):
        """Arithmetic of the pair forces: "auto" (default: float64 for the waves whose bodies sit densely enough that
        G rho dt^2 > tau, fp32 elsewhere), "f32", "f64" (the reference's own arithmetic, nbody/simulation.py:246-268).
        The accepted (body, node) sets are the reference's in every mode."""
        _nat.check(self._lib.nbmi_set_force_precision(self._h, self.FORCE_PRECISION[mode], float(tau)),
                   "nbmi_set_force_precision")

    def tree_stats(self, depth=True):
        """num_nodes as build_octree returns it, max depth, root half size (compute_bounds).
        depth=False skips the reduction kernel behind max_depth (one small D2H copy only)."""
        nn, md, b = C.c_int64(0), C.c_int32(0), C.c_double(0)
        _nat.check(self._lib.nbmi_tree_stats(self._h, C.addressof(nn), C.addressof(md) if depth else None,
                                             C.addressof(b)), "nbmi_tree_stats")
        out = dict(num_nodes=int(nn.value), bounds=float(b.value))
        if depth:
            out["max_depth"] = int(md.value)
        return out

    def morton_keys(self):
        """(key_hi, key_lo) uint64 per body, caller's order, for the last built tree."""
        hi = np.empty(self.n, dtype=np.uint64)
        lo = np.empty(self.n, dtype=np.uint64)
        _nat.check(self._lib.nbmi_get_keys(self._h, _nat.ptr(hi), _nat.ptr(lo)), "nbmi_get_keys")
        return hi, lo

    def sort_keys(self):
        """The keys the device sorts by (octant digits relabelled along the Hilbert curve), caller's order."""
        hi = np.empty(self.n, dtype=np.uint64)
        lo = np.empty(self.n, dtype=np.uint64)
        _nat.check(self._lib.nbmi_get_sort_keys(self._h, _nat.ptr(hi), _nat.ptr(lo)), "nbmi_get_sort_keys")
        return hi, lo

    def cells(self):
        """(level, key) of every node of the last built tree."""
        nn = self.tree_stats()["num_nodes"]
        level = np.empty(nn, dtype=np.int32)
        key = np.empty(nn, dtype=np.uint64)
        _nat.check(self._lib.nbmi_get_cells(self._h, _nat.ptr(level), _nat.ptr(key), nn), "nbmi_get_cells")
        return level, key

    def walk_counters(self):
        out = np.zeros(17, dtype=np.int64)
        _nat.check(self._lib.nbmi_walk_counters(self._h, _nat.ptr(out)), "nbmi_walk_counters")
        return dict(wave_visits=int(out[0]), lane_visits=int(out[1]), lane_accepts=int(out[2]),
                    window_misses={8 << w: int(out[3 + w]) for w in range(4)}, jumps=int(out[7]),
                    xcd_visits=[int(v) for v in out[8:16]], band_visits=int(out[16]))

    def key_order(self):
        """Body indices along the sort-key order of the last built tree (octree DFS, the eight children of a cell
        in Hilbert-curve order; see sort_keys)."""
        out = np.empty(self.n, dtype=np.int32)
        _nat.check(self._lib.nbmi_get_order(self._h, _nat.ptr(out)), "nbmi_get_order")
        return out


class HIPOwnerSimulation(HIPBarnesHutSimulation):
    """Owner-mode handle of the multi-GPU stage 2 (include/nbmi.h, nbmi_create_owner): the bodies of one
    octant-key range, their own octree inside the global root cube, plus received locally essential trees.
    Driven by nbody/sharded.py::LetBarnesHut; buffers are device pointers."""

    def __init__(self, positions, velocities, masses, global_ids, capacity, let_capacity, world, rank, G, softening,
                 damping, theta=0.5, device=None):
        lib = _nat.load()
        pos = _as_f64(positions, (3,))
        vel = _as_f64(velocities, (3,))
        m = _as_f64(masses, ())
        ids = np.ascontiguousarray(global_ids, dtype=np.int32)
        if not (len(pos) == len(vel) == len(m) == len(ids)):
            raise ValueError("positions, velocities, masses and ids must have the same length")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, _nat.device_count())
        self.G, self.softening, self.damping, self.theta = float(G), float(softening), float(damping), float(theta)
        self.device = int(device)
        self.world, self.rank = int(world), int(rank)
        self.capacity, self.let_capacity = int(capacity), int(let_capacity)
        self._lib = lib
        self._h = lib.nbmi_create_owner(len(pos), _nat.ptr(pos), _nat.ptr(vel), _nat.ptr(m), _nat.ptr(ids), self.capacity,
                                        self.let_capacity, self.world, self.rank, self.G, self.softening, self.damping,
                                        self.theta, self.device)
        if not self._h:
            raise RuntimeError(f"nbmi_create_owner failed: {_nat.last_error()}")
        print(f"[HIP] rank {rank}/{world}: owner of {len(pos):,} bodies (capacity {capacity:,}) on device {self.device}")

    @property
    def n(self):
        return int(self._lib.nbmi_owner_count(self._h)) if self._h else 0

    def ids(self):
        out = np.empty(self.n, dtype=np.int32)
        _nat.check(self._lib.nbmi_owner_get_ids(self._h, _nat.ptr(out)), "nbmi_owner_get_ids")
        return out

    def owner_maxabs(self, dev_maxabs):
        _nat.check(self._lib.nbmi_owner_maxabs(self._h, int(dev_maxabs)), "nbmi_owner_maxabs")

    def owner_sample(self, dev_maxabs, dev_samples, nsamples, nvalid=0):
        """`nvalid` of the `nsamples` slots get a key sample (0: all of them)."""
        _nat.check(self._lib.nbmi_owner_sample(self._h, int(dev_maxabs), int(dev_samples), int(nsamples), int(nvalid)),
                   "nbmi_owner_sample")

    def owner_partition(self, dev_all_samples, total, dev_send_rows):
        counts = np.zeros(self.world, dtype=np.int64)
        _nat.check(self._lib.nbmi_owner_partition(self._h, int(dev_all_samples), int(total), int(dev_send_rows),
                                                  _nat.ptr(counts)), "nbmi_owner_partition")
        return counts

    def owner_adopt(self, dev_recv_rows, n_new, dev_maxabs, dev_bbox, dev_chain=0):
        _nat.check(self._lib.nbmi_owner_adopt(self._h, int(dev_recv_rows), int(n_new), int(dev_maxabs), int(dev_bbox), int(dev_chain)),
                   "nbmi_owner_adopt")

    def chain_doubles(self):
        """float64 words of a rank's boundary table (nbmi_owner_adopt writes it, all ranks' tables go to owner_export_let)."""
        return int(self._lib.nbmi_owner_chain_doubles())

    def owner_export_let(self, dev_boxes, dev_chains, dev_let):
        """Rows for every destination rank (packed in rank order in `dev_let`); returns the counts."""
        counts = np.zeros(self.world, dtype=np.int64)
        _nat.check(self._lib.nbmi_owner_export_let(self._h, int(dev_boxes), int(dev_chains), int(dev_let), _nat.ptr(counts)),
                   "nbmi_owner_export_let")
        return counts

    def owner_set_dt(self, dt):
        """dt of the step about to be exchanged: force precision "auto" is decided while owner_adopt builds the tree."""
        _nat.check(self._lib.nbmi_owner_set_dt(self._h, float(dt)), "nbmi_owner_set_dt")

    def let_row_bytes(self):
        return int(self._lib.nbmi_owner_let_row_bytes())

    def owner_step_facts(self):
        """After owner_export_let: int64[4] = (waves asking for float64 forces, waves, tree rows that fit in front of /
        behind the own piece of the walk array)."""
        out = np.zeros(4, dtype=np.int64)
        _nat.check(self._lib.nbmi_owner_step_facts(self._h, _nat.ptr(out)), "nbmi_owner_step_facts")
        return out

    def owner_set_all64(self, verdict):
        """The ranks' common decision for the next owner_step: True / False = every wave float64 / the waves decide;
        None = this rank's own rule."""
        _nat.check(self._lib.nbmi_owner_set_all64(self._h, -1 if verdict is None else int(bool(verdict))), "nbmi_owner_set_all64")

    def owner_step(self, dev_lets, counts, dt):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        _nat.check(self._lib.nbmi_owner_step(self._h, int(dev_lets), _nat.ptr(counts), float(dt)), "nbmi_owner_step")


class HIPDirectSimulation(_HIPSimulation):
    """All-pairs O(N^2) forces, LDS tiled (reference CUDASimulation, nbody/gpu_backend.py:336-409;
    no theta)."""

    _method = METHOD_DIRECT

    def __init__(self, positions, velocities, masses, G, softening, damping, device=None):
        super().__init__(positions, velocities, masses, G, softening, damping, theta=0.0, device=device)


# Reference thresholds (:618-620) exist because its GPU paths are O(N^2); the HIP Barnes-Hut
# backend is O(N log N) like the reference's Metal one, so it takes every size.
HIP_BH_THRESHOLD = 100_000_000  # = kMaxBodies of libnbmi.so (node links are 32-bit byte offsets): beyond it the factory returns None


def create_gpu_simulation(positions: np.ndarray, velocities: np.ndarray, masses: np.ndarray, G: float,
                          softening: float, damping: float, theta: float = 0.5, force_gpu: bool = False,
                          method: Optional[str] = None):
    """Reference signature (:623-625) plus ``method`` ("barnes_hut" default, or "direct").

    Returns a backend object, or None if the HIP backend is not available / not selected."""
    backend, _info = get_backend()
    n = len(positions)
    if backend != Backend.HIP:
        return None
    method = method or os.environ.get("NBMI_METHOD", "barnes_hut")
    if method == "direct":
        return HIPDirectSimulation(positions, velocities, masses, G, softening, damping)
    if n <= HIP_BH_THRESHOLD or force_gpu:
        return HIPBarnesHutSimulation(positions, velocities, masses, G, softening, damping, theta)
    print(f"[GPU] {n:,} bodies exceeds the HIP Barnes-Hut limit ({HIP_BH_THRESHOLD:,})")
    return None
