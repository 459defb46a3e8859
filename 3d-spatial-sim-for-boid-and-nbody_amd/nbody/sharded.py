"""Multi-GPU Barnes-Hut: one process per GPU, bodies sharded by octant-key range.

The reference is single-device; this is new design, in two forms, both bit-identical
to the 1-GPU result (the force on a body depends only on the global tree and on its own state):

* **row exchange** (``ShardedBarnesHut``, stage 1, the default of ``create_sharded_simulation``):
  every rank holds all N bodies, sorts and builds over all of them, integrates one contiguous
  range of sorted ranks (always a compact region, whatever the bodies did), and all-gathers the
  updated rows {x,y,z,vx,vy,vz,m,id} (64 B per body).
* **run exchange** (``RunExchangeBarnesHut``, experimental): a rank OWNS a fixed set of bodies
  (initially one contiguous range of the key order) and keeps their float64 state to itself.  Per
  step: all-reduce MAX of one double (the root cube), local keys + local sort, all-gather of the
  key-sorted runs (32 B per body: two key words + fp32 x,y,z,G m), merge of the ``world`` sorted
  runs, octree over the whole system, walk + integrate the owned bodies.  Half the bytes on the
  wire and no whole-system sort, but measured it only pays with body
  MIGRATION: a body that leaves its owner's region keeps its owner, lands in a wave with other
  emigrants from all over the system, and that one wave then walks ~64 bodies' worth of distinct
  paths serially (walk 1.7 -> 5-7 ms after a single step at 4 x 1 M).  It is the stepping stone to
  the locally-essential-tree exchange, not the production path.

Collectives are ``torch.distributed`` calls on device buffers (backend "nccl" = RCCL over xGMI).

``ShardedBarnesHut`` is written against a small shard-engine interface so the collective logic
can be exercised on CPU (gloo) with a stand-in engine:
    set_shard(begin, end) / step(dt) / export_rows(out) / import_rows(full, n_rows)
``HipShardEngine`` is the real one (device pointers into libnbmi.so).
"""
import os

import numpy as np

ROW = 8  # doubles per packed body row (include/nbmi.h nbmi_export_shard)


def shard_bounds(n: int, world: int, rank: int):
    """Equal-count contiguous ranges of sorted ranks; `per` is the padded all-gather chunk."""
    per = (n + world - 1) // world
    begin = min(n, rank * per)
    end = min(n, begin + per)
    return per, begin, end


class HipShardEngine:
    """Shard engine on top of HIPBarnesHutSimulation; rows travel as torch CUDA tensors."""

    def __init__(self, positions, velocities, masses, G, softening, damping, theta, device, method="barnes_hut"):
        import torch
        from .gpu_backend import HIPBarnesHutSimulation, HIPDirectSimulation
        self.torch = torch
        self.device = torch.device("cuda", device)
        if method == "direct":  # all-pairs: rows sharded by body index, same exchange
            self.sim = HIPDirectSimulation(positions, velocities, masses, G, softening, damping, device=device)
        else:
            self.sim = HIPBarnesHutSimulation(positions, velocities, masses, G, softening, damping, theta, device=device)
        self.n = self.sim.n
        self.stream = None

    def new_rows(self, rows):
        t = self.torch.zeros((rows, ROW), dtype=self.torch.float64, device=self.device)
        # the fill ran on torch's stream; the library's stream is non-blocking, so nothing else orders it
        # before the first pack kernel / collective that touches the buffer
        self.torch.cuda.synchronize(self.device)
        return t

    def set_shard(self, begin, end):
        self.sim.set_shard(begin, end)

    def step(self, dt):
        self.sim.step(dt)

    def share_stream(self):
        """Make the library's own HIP stream torch's current stream for the exchange: the
        collective is then ordered after the pack kernel and before the unpack kernel by the
        stream itself, and a step needs no host synchronisation.  Returns the torch stream."""
        if self.stream is None:
            stream = self.torch.cuda.ExternalStream(self.sim.stream_handle(), device=self.device)
            self.sim.set_exchange_sync(False)
            self.stream = stream  # only now: import_rows() skips its host synchronisation when this is set
        return self.stream

    def export_rows(self, out):
        self.sim.export_shard(out.data_ptr())  # synchronises the library stream unless it is shared

    def import_rows(self, full, n_rows):
        if self.stream is None:
            self.torch.cuda.current_stream(self.device).synchronize()  # collective finished
        self.sim.import_ranks(full.data_ptr(), 0, n_rows)


class ShardedBarnesHut:
    """step()/get_* over `world` ranks; every rank ends each step with the full updated state."""

    def __init__(self, engine, n, rank, world, dist=None):
        self.engine, self.n, self.rank, self.world = engine, n, rank, world
        self.dist = dist
        self.per, self.begin, self.end = shard_bounds(n, world, rank)
        engine.set_shard(self.begin, self.end)
        self.mine = engine.new_rows(self.per)
        self.full = engine.new_rows(self.per * world)
        # The exchange synchronises on the host by default.  NBMI_EXCHANGE_SYNC=0 opts in to running the
        # RCCL collective on the library's own stream (no host synchronisation inside a step; measured
        # 1.67 -> 1.59 ms/step with one rank): it has only ever run with ONE rank, so it stays opt-in until a
        # run on >= 2 GPUs has matched the single-handle result bit for bit.
        self.shared = None
        if dist is not None and hasattr(engine, "share_stream") and dist.get_backend() == "nccl" \
                and os.environ.get("NBMI_EXCHANGE_SYNC", "1") == "0":
            try:
                self.shared = engine.share_stream()
            except Exception as ex:  # noqa: BLE001 - keep the host-synchronised exchange rather than fail
                import sys
                print(f"[sharded] stream-ordered exchange unavailable ({ex}); synchronising on the host", file=sys.stderr)
                self.shared = None
                engine.stream = None
                with __import__("contextlib").suppress(Exception):
                    engine.sim.set_exchange_sync(True)

    def step(self, dt, substeps=1):
        if self.dist is not None and self.shared is not None:
            # everything - kernels, pack, collective, unpack - is enqueued in order on the library's stream
            with self.engine.torch.cuda.stream(self.shared):
                for _ in range(substeps):
                    self.engine.step(dt)
                    self.engine.export_rows(self.mine)
                    self.dist.all_gather_into_tensor(self.full, self.mine)
                    self.engine.import_rows(self.full, self.n)
            return
        for _ in range(substeps):
            self.engine.step(dt)
            if self.dist is None:  # one rank, no process group: nothing to exchange
                continue
            self.engine.export_rows(self.mine)
            self.dist.all_gather_into_tensor(self.full, self.mine)
            self.engine.import_rows(self.full, self.n)


def create_sharded_simulation(positions, velocities, masses, G, softening, damping, theta=0.5, mode="rows",
                              method="barnes_hut"):
    """Build the multi-GPU stepper from the torch.distributed environment (RANK/LOCAL_RANK/
    WORLD_SIZE).  Every rank passes the same full arrays.  mode: "rows" (stage 1, replicated state)
    or "runs" (experimental run exchange with fixed ownership); method "direct" shards the all-pairs
    kernel by body index through the same row exchange."""
    import os
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    import torch
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
    if mode == "rows" or method == "direct":
        eng = HipShardEngine(positions, velocities, masses, G, softening, damping, theta, local, method=method)
        # (a process group of one rank still runs the collective: used to smoke-test RCCL on a 1-GPU box)
        return ShardedBarnesHut(eng, len(positions), rank, world, dist if dist.is_initialized() else None)
    if mode != "runs":
        raise ValueError(f"unknown sharding mode {mode!r}")
    eng = HipRunEngine(positions, velocities, masses, G, softening, damping, theta, local, rank, world)
    return RunExchangeBarnesHut(eng, rank, world, DistComm(dist) if world > 1 else None)


class HipRunEngine:
    """Owner engine of the run exchange: this rank's bodies in one libnbmi handle."""

    def __init__(self, positions, velocities, masses, G, softening, damping, theta, device, rank, world):
        import torch
        from .gpu_backend import HIPBarnesHutSimulation
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.n_total = len(positions)
        self.per, begin, end = shard_bounds(self.n_total, world, rank)
        if world == 1:
            self.ids = np.arange(self.n_total, dtype=np.int64)
        else:
            # initial owners: contiguous ranges of the key order (compact wave groups)
            full = HIPBarnesHutSimulation(positions, velocities, masses, G, softening, damping, theta, device=device)
            full.build_tree()
            self.ids = full.key_order()[begin:end].astype(np.int64)
            full.close()
        self.sim = HIPBarnesHutSimulation(np.asarray(positions)[self.ids], np.asarray(velocities)[self.ids],
                                          np.asarray(masses)[self.ids], G, softening, damping, theta, device=device)
        self.n = self.sim.n
        self.world = world
        self.sim.exchange_enable(self.n_total, world, self.per)

    def new_maxabs(self):
        return self.torch.zeros(1, dtype=self.torch.float64, device=self.device)

    def new_run(self, rows):
        return self.torch.zeros((rows, 4), dtype=self.torch.int64, device=self.device)  # 32-byte records

    def local_maxabs(self, out):
        self.sim.exchange_maxabs(out.data_ptr())  # synchronises the library stream

    def export_run(self, maxabs, out):
        self.torch.cuda.current_stream(self.device).synchronize()  # all-reduce finished
        self.sim.exchange_export(maxabs.data_ptr(), out.data_ptr(), out.shape[0])

    def step_runs(self, full, dt):
        self.torch.cuda.current_stream(self.device).synchronize()  # all-gather finished
        self.sim.exchange_step(full.data_ptr(), self.world, full.shape[0] // self.world, dt)

    def owned_state(self):
        """(global ids, positions f64, velocities f64) of the owned bodies."""
        return self.ids, self.sim.get_positions_f64(), self.sim.get_velocities()


class RunExchangeBarnesHut:
    """step() over `world` ranks with fixed body ownership; see the module docstring.
    `comm` needs all_reduce_max(tensor) and all_gather(full, mine) (default: torch.distributed)."""

    def __init__(self, engine, rank, world, comm=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.comm = comm
        self.n = engine.n_total
        self.per = engine.per
        self.maxabs = engine.new_maxabs()
        self.mine = engine.new_run(self.per)
        self.full = engine.new_run(self.per * world) if world > 1 else self.mine

    def step(self, dt, substeps=1):
        e = self.engine
        for _ in range(substeps):
            e.local_maxabs(self.maxabs)
            if self.world > 1:
                self.comm.all_reduce_max(self.maxabs)
            e.export_run(self.maxabs, self.mine)
            if self.world > 1:
                self.comm.all_gather(self.full, self.mine)
            e.step_runs(self.full, dt)

    def gather_state(self):
        """Full (positions, velocities) float64 in the caller's original order, on every rank."""
        ids, pos, vel = self.engine.owned_state()
        if self.world == 1:
            out_p, out_v = np.empty((self.n, 3)), np.empty((self.n, 3))
            out_p[ids], out_v[ids] = pos, vel
            return out_p, out_v
        import torch
        mine = torch.full((self.per, 7), -1.0, dtype=torch.float64)
        mine[: len(ids), 0] = torch.from_numpy(ids.astype(np.float64))
        mine[: len(ids), 1:4] = torch.from_numpy(pos)
        mine[: len(ids), 4:7] = torch.from_numpy(vel)
        dev = self.full.device
        mine = mine.to(dev)
        full = torch.empty((self.per * self.world, 7), dtype=torch.float64, device=dev)
        self.comm.all_gather(full, mine)
        rows = full.cpu().numpy()
        rows = rows[rows[:, 0] >= 0]
        out_p, out_v = np.empty((self.n, 3)), np.empty((self.n, 3))
        gid = rows[:, 0].astype(np.int64)
        out_p[gid], out_v[gid] = rows[:, 1:4], rows[:, 4:7]
        return out_p, out_v


class DistComm:
    """The two collectives of the run exchange on torch.distributed."""

    def __init__(self, dist):
        self.dist = dist

    def all_reduce_max(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)

    def all_gather(self, full, mine):
        self.dist.all_gather_into_tensor(full, mine)


def unpack_rows(rows: np.ndarray):
    """(positions, velocities, masses, ids) from packed rows."""
    return rows[:, 0:3], rows[:, 3:6], rows[:, 6], rows[:, 7].astype(np.int64)
