"""Multi-GPU Barnes-Hut: one process per GPU, bodies sharded by octant-key range.

The reference is single-device; this is new design.  Stage 1 ("replicate the tree",
bit-identical to the 1-GPU result): every rank holds all N bodies and builds the full octree
(the non-scaling part), walks + integrates only its contiguous range of key-sorted ranks, and one
all-gather per step (RCCL over xGMI: ``torch.distributed`` backend "nccl") returns everybody's
updated rows {x,y,z,vx,vy,vz,m,id} (64 B per body).  The force on a body depends only on the
global tree and on its own state, so the result does not depend on the world size.

``ShardedBarnesHut`` is written against a small shard-engine interface so the collective logic
can be exercised on CPU (gloo) with a stand-in engine:
    set_shard(begin, end) / step(dt) / export_rows(out) / import_rows(full, n_rows)
``HipShardEngine`` is the real one (device pointers into libnbmi.so).
"""
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

    def __init__(self, positions, velocities, masses, G, softening, damping, theta, device):
        import torch
        from .gpu_backend import HIPBarnesHutSimulation
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.sim = HIPBarnesHutSimulation(positions, velocities, masses, G, softening, damping, theta, device=device)
        self.n = self.sim.n

    def new_rows(self, rows):
        return self.torch.zeros((rows, ROW), dtype=self.torch.float64, device=self.device)

    def set_shard(self, begin, end):
        self.sim.set_shard(begin, end)

    def step(self, dt):
        self.sim.step(dt)

    def export_rows(self, out):
        self.sim.export_shard(out.data_ptr())  # synchronises the library stream

    def import_rows(self, full, n_rows):
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

    def step(self, dt, substeps=1):
        for _ in range(substeps):
            self.engine.step(dt)
            if self.world == 1:
                continue
            self.engine.export_rows(self.mine)
            self.dist.all_gather_into_tensor(self.full, self.mine)
            self.engine.import_rows(self.full, self.n)


def create_sharded_simulation(positions, velocities, masses, G, softening, damping, theta=0.5):
    """Build a ShardedBarnesHut from the torch.distributed environment (RANK/LOCAL_RANK/WORLD_SIZE).
    Every rank passes the same full arrays."""
    import os
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    import torch
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
    eng = HipShardEngine(positions, velocities, masses, G, softening, damping, theta, local)
    return ShardedBarnesHut(eng, len(positions), rank, world, dist if world > 1 else None)


def unpack_rows(rows: np.ndarray):
    """(positions, velocities, masses, ids) from packed rows."""
    return rows[:, 0:3], rows[:, 3:6], rows[:, 6], rows[:, 7].astype(np.int64)
