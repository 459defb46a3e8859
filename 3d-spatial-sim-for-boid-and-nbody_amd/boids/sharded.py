"""Multi-GPU boids: x-slabs of whole cell planes with a one-cell halo, one process per GPU.

The reference is single-process (boids/flock.py).  A boid only interacts within the perception radius = one
grid cell (flock.py:478-481), so a rank that owns the cell planes [p0, p1) needs, besides its own boids, the
boids of the two adjacent planes.  Per step every rank hands its two neighbours the boids that lie within one
cell of the shared face - owned boids that have just crossed the face included, which is how boids migrate
(max_speed * dt is far below one cell) - then runs the reference's Flock.update on owned + ghost boids with
the ghosts read-only (bdmi_slab_* in include/bdmi.h).  Every owned boid sees exactly the candidates it would
see on one GPU; only the float64 summation order differs.

``SlabFlock.step`` is written against an engine interface (op_export / op_import / op_step and torch buffers)
and a communicator with all_to_all_counts / all_to_all_rows (nbody.sharded.DistComm on torch.distributed;
tests play the ranks with threads on one GPU).
"""
import ctypes as C

import numpy as np

import nbmi_native as _nat

ROW = 10  # doubles per boid row: position, velocity, colour, global id


def slab_planes(grid_dim: int, world: int):
    """Cell-plane boundaries of the `world` slabs: rank r owns planes [b[r], b[r+1])."""
    b = [round(r * grid_dim / world) for r in range(world + 1)]
    if any(b[r + 1] - b[r] < 2 for r in range(world)):
        raise ValueError(f"{world} slabs of a {grid_dim}-plane grid would be narrower than two cells")
    return b


class HipSlabEngine:
    """One rank's slab in a bdmi slab-mode handle; exchange buffers are torch CUDA tensors."""

    def __init__(self, positions, velocities, colors, params, rank, world, device=0, headroom=1.5):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.rank, self.world = rank, world
        self.params = np.ascontiguousarray(params, dtype=np.float64)
        bounds, cell = float(self.params[0]), float(self.params[5])
        dim = int(np.ceil(bounds * 2 / cell)) + 2
        offset = bounds + cell
        planes = slab_planes(dim, world)
        self.x_lo, self.x_hi = planes[rank] * cell - offset, planes[rank + 1] * cell - offset
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        self.n_total = len(pos)
        lo = -np.inf if rank == 0 else self.x_lo
        hi = np.inf if rank == world - 1 else self.x_hi
        ids = np.nonzero((pos[:, 0] >= lo) & (pos[:, 0] < hi))[0].astype(np.int32)
        self.cap = int(headroom * max(len(ids), self.n_total // world)) + 4096
        lib = _nat.load()
        self._lib = lib
        p = np.ascontiguousarray(pos[ids])
        v = np.ascontiguousarray(np.asarray(velocities, dtype=np.float64)[ids])
        c = np.ascontiguousarray(np.asarray(colors, dtype=np.float64)[ids])
        self._h = lib.bdmi_create_slab(len(ids), _nat.ptr(p), _nat.ptr(v), _nat.ptr(c), _nat.ptr(ids), self.cap,
                                       _nat.ptr(self.params), self.x_lo, self.x_hi, int(rank > 0), int(rank < world - 1),
                                       device)
        if not self._h:
            raise RuntimeError(f"bdmi_create_slab failed: {_nat.last_error()}")
        z = lambda rows: torch.zeros((rows, ROW), dtype=torch.float64, device=self.device)  # noqa: E731
        self.left, self.right = z(self.cap), z(self.cap)
        self.send, self.recv = z(2 * self.cap), z(2 * self.cap)
        torch.cuda.synchronize(self.device)
        self.sent_rows = 0

    def op_export(self):
        """Rows for the left and right neighbour, packed in rank order into `send`; returns the send counts."""
        nl, nr = C.c_int64(0), C.c_int64(0)
        _nat.check(self._lib.bdmi_slab_export(self._h, self.left.data_ptr(), C.addressof(nl), self.right.data_ptr(),
                                              C.addressof(nr)), "bdmi_slab_export")
        _nat.check(self._lib.bdmi_sync(self._h), "bdmi_sync")
        nl, nr = int(nl.value), int(nr.value)
        counts = np.zeros(self.world, dtype=np.int64)
        if nl:
            counts[self.rank - 1] = nl
            self.send[:nl].copy_(self.left[:nl])
        if nr:
            counts[self.rank + 1] = nr
            self.send[nl:nl + nr].copy_(self.right[:nr])
        self.torch.cuda.current_stream(self.device).synchronize()
        self.sent_rows = nl + nr
        return counts

    def op_import(self, count):
        _nat.check(self._lib.bdmi_slab_import(self._h, self.recv.data_ptr(), int(count)), "bdmi_slab_import")

    def op_step(self, dt):
        _nat.check(self._lib.bdmi_step(self._h, float(dt), 1), "bdmi_step")

    def wait(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def owned_rows(self):
        """(count, 10) float64 rows {p, v, c, global id} of the owned boids."""
        out = np.empty((self.cap, ROW))
        cnt = C.c_int64(0)
        _nat.check(self._lib.bdmi_slab_get(self._h, _nat.ptr(out), self.cap, C.addressof(cnt)), "bdmi_slab_get")
        return out[: int(cnt.value)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bdmi_destroy(self._h)
            self._h = None


class SlabFlock:
    """Flock.update over `world` slabs: one neighbour exchange, then the reference's update on every slab."""

    def __init__(self, engine, rank, world, comm=None):
        self.engine, self.rank, self.world, self.comm = engine, rank, world, comm
        if world > 1 and comm is None:
            raise ValueError("SlabFlock over more than one rank needs a communicator")

    def step(self, dt, substeps=1):
        e = self.engine
        for _ in range(substeps):
            send_counts = e.op_export()
            if self.world > 1:
                recv_counts = self.comm.all_to_all_counts(send_counts)
                self.comm.all_to_all_rows(e.recv, e.send, recv_counts, send_counts)
                e.wait()
                e.op_import(int(recv_counts.sum()))
            e.op_step(dt)

    def gather_state(self, n_total):
        """(positions, velocities, colors) of ALL boids in global-id order, on every rank (verification)."""
        import torch
        rows = self.engine.owned_rows()
        if self.world == 1:
            full = rows
        else:
            cap = n_total  # the same on every rank (the engines' own capacities differ)
            mine = torch.full((cap, ROW), -1.0, dtype=torch.float64)
            mine[: len(rows)] = torch.from_numpy(rows)
            mine = mine.to(self.engine.send.device)
            allr = torch.empty((cap * self.world, ROW), dtype=torch.float64, device=mine.device)
            self.comm.all_gather(allr, mine)
            self.engine.wait()
            full = allr.cpu().numpy()
            full = full[full[:, 9] >= 0]
        gid = full[:, 9].astype(np.int64)
        assert len(gid) == n_total and len(np.unique(gid)) == n_total, "every boid must have exactly one owner"
        out = np.empty((n_total, 9))
        out[gid] = full[:, :9]
        return out[:, 0:3], out[:, 3:6], out[:, 6:9]
