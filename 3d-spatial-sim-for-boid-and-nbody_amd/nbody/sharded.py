"""Multi-GPU Barnes-Hut: one process per GPU, bodies sharded by key range (octant-path keys, children of a cell
ordered along the Hilbert curve: a range of the key order is a compact piece of space).

The reference is single-device; this is new design, in two forms:

* **row exchange** (``ShardedBarnesHut``, stage 1, ``mode="rows"``): every rank holds all N bodies, sorts
  and builds over all of them, integrates one contiguous range of sorted ranks, and all-gathers the
  updated rows {x,y,z,vx,vy,vz,m,id} (64 B per body).  Bit-identical to the 1-GPU result for any world
  size (the force on a body depends only on the global tree and on its own state); per-rank sort, build
  and memory grow with the whole system - it is the exact reference mode, not the scaling one.
* **locally essential trees** (``LetBarnesHut``, stage 2, ``mode="let"`` - the form BASELINE's north_star
  names): a rank OWNS the bodies of one key range.  Per step: all-reduce MAX of one double (the root cube
  of the whole system), keys, all-gather of a few hundred key samples -> splitters at equal quantiles,
  all-to-all-v of the rows that crossed a splitter (body migration: a rank's bodies always form one compact
  key range, so its waves stay compact), local sort + octree of the owned bodies inside the global cube,
  all-gather of the ranks' bounding boxes (tight boxes of the cells of each rank's own tree), pruning of
  the own tree against EACH other rank's boxes (the reference's opening test at the box's nearest point,
  conservative by 1e-9), all-to-all-v of the pruned trees (48 B per node, float64 moments: the receiver rebuilds the fp32 and
  the float64 walk record from them; a rank only receives what its own bodies can open), walk over own +
  received trees with the handle's force precision ("auto" by default, as on one GPU).
  Per-rank sort / build / state no longer grow with the world size; only the received trees do.  The
  ranks' trees are pieces of ONE global octree (cells that span ranks carry global moments; each rank's
  boundary table travels with its boxes): every body visits the 1-GPU run's accepted nodes in the 1-GPU order.

Collectives are ``torch.distributed`` calls on device buffers (backend "nccl" = RCCL over xGMI), behind a
small comm interface so that tests can play the ranks with threads on one GPU.

``ShardedBarnesHut`` is written against a small shard-engine interface so the collective logic
can be exercised on CPU (gloo) with a stand-in engine:
    set_shard(begin, end) / step(dt) / export_rows(out) / import_rows(full, n_rows)
``HipShardEngine`` is the real one (device pointers into libnbmi.so).
"""
import os

import numpy as np

ROW = 8  # doubles per packed body row (include/nbmi.h nbmi_export_shard)
ALL64_ENTER, ALL64_LEAVE = 0.333, 0.25  # csrc/nbmi.hip kAll64Enter / kAll64Leave (per mille there)


def shard_bounds(n: int, world: int, rank: int):
    """Equal-count contiguous ranges of sorted ranks; `per` is the padded all-gather chunk.  A range starts at a
    multiple of 64 ranks: a wave's 64 bodies are then the same ones however the ranks are sharded, and so is the
    place where the walk cuts the node array between its two cursors (large systems cut at the wave's own
    leaves) - the condition for results that do not depend on the sharding bit for bit."""
    per = (n + world - 1) // world
    per = (per + 63) // 64 * 64
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
    WORLD_SIZE).  Every rank passes the same full arrays.  mode: "rows" (stage 1, replicated state and
    tree, bit-exact) or "let" (stage 2, owned key ranges + locally essential trees); method "direct"
    shards the all-pairs kernel by body index through the row exchange."""
    import os
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
    if mode == "rows" or method == "direct":
        eng = HipShardEngine(positions, velocities, masses, G, softening, damping, theta, local, method=method)
        # (a process group of one rank still runs the collective: used to smoke-test RCCL on a 1-GPU box)
        return ShardedBarnesHut(eng, len(positions), rank, world, dist if dist.is_initialized() else None)
    if mode != "let":
        raise ValueError(f"unknown sharding mode {mode!r}")
    eng = HipLetEngine(positions, velocities, masses, G, softening, damping, theta, local, rank, world)
    return LetBarnesHut(eng, rank, world, DistComm(dist, eng.device) if dist.is_initialized() else None)


def let_capacities(n_total, world):
    """(body capacity, rows for exchanged trees) of a rank: 25 % head room over the equal share (the splitters
    re-balance every step, so the share only drifts by sampling noise) and, for the trees a rank sends to / receives
    from all the others together, as many rows as its own tree has (~1.5 per body; measured at 1 M bodies per
    rank and 8 ranks: a third of that)."""
    share = (n_total + world - 1) // world
    cap = int(share * 1.25) + 4096
    let = 0 if world == 1 else int(1.5 * share) + 65536
    return cap, let


class HipLetEngine:
    """This rank's bodies in one owner-mode libnbmi handle; exchange buffers are torch CUDA tensors."""

    SAMPLES = 256        # key samples per rank for the splitters
    LET_ROW_BYTES = 48   # float64 {cx, cy, cz, G m}, float s2t, uint next, float64 half size (nbmi_owner_let_row_bytes)

    def __init__(self, positions, velocities, masses, G, softening, damping, theta, device, rank, world):
        import torch
        from .gpu_backend import HIPBarnesHutSimulation, HIPOwnerSimulation
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.rank, self.world = rank, world
        self.n_total = n = len(positions)
        per, begin, end = shard_bounds(n, world, rank)
        if world == 1:
            ids = np.arange(n, dtype=np.int32)
        else:
            # initial owners: contiguous ranges of the key order (the first step re-balances anyway)
            full = HIPBarnesHutSimulation(positions, velocities, masses, G, softening, damping, theta, device=device)
            full.build_tree()
            ids = full.key_order()[begin:end].astype(np.int32)
            full.close()
        cap, let = let_capacities(n, world)
        self.cap, self.let_rows = cap, let
        self.SAMPLES = min(self.SAMPLES, 4096 // world)  # the library sorts world x SAMPLES keys in one workgroup
        self.sim = HIPOwnerSimulation(np.asarray(positions)[ids], np.asarray(velocities)[ids], np.asarray(masses)[ids], ids,
                                      cap, let, world, rank, G, softening, damping, theta, device=device)
        f64, i64 = torch.float64, torch.int64
        z = lambda *shape, dtype=f64: torch.zeros(shape, dtype=dtype, device=self.device)  # noqa: E731
        self.maxabs = z(1)
        self.samples = z(self.SAMPLES, dtype=i64)
        self.all_samples = z(world * self.SAMPLES, dtype=i64)
        self.send_rows = z(cap, ROW)
        self.recv_rows = z(cap, ROW)
        nbox = 6 * int(self.sim._lib.nbmi_owner_boxes_per_rank())  # several boxes per rank: a key range is not a box
        self.bbox = z(nbox)
        self.boxes = z(world * nbox)
        # what the other ranks need to know about this rank's two ends to cut ONE global octree into the ranks' pieces
        nchain = self.sim.chain_doubles()
        self.chain = z(nchain)
        self.chains = z(world * nchain)
        # trees travel as 48-byte float64 rows (both walk records are rebuilt from them), one packed segment per rank
        self.LET_ROW_BYTES = self.sim.let_row_bytes()
        self.let_send = torch.zeros((max(let, 1), self.LET_ROW_BYTES), dtype=torch.uint8, device=self.device)
        self.let_recv = torch.zeros((max(let, 1), self.LET_ROW_BYTES), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)  # the fills ran on torch's stream, the library has its own
        # [r3] Collectives are enqueued on the LIBRARY's stream (torch sees it as an external stream): kernels, packs,
        # collectives and unpacks order themselves, and the only host waits of a step are the two that bring the
        # variable exchange sizes to the host (migration counts, tree counts).  NBMI_EXCHANGE_SYNC=1: the round-2
        # behaviour (a host synchronisation after every phase and collective).
        self.stream = None
        if os.environ.get("NBMI_EXCHANGE_SYNC", "0") != "1":
            self.stream = torch.cuda.ExternalStream(self.sim.stream_handle(), device=self.device)
            self.sim.set_exchange_sync(False)
        self.wire_bytes = 0  # bytes this rank sent in the last step (rows + tree + small collectives)
        self.migrated = 0    # bodies this rank handed to other ranks in the last step
        self.let_counts = np.zeros(world, dtype=np.int64)

    # ---- the six phases of a step (LetBarnesHut puts the collectives between them) ------------------
    def op_begin(self, dt):
        """dt of the step that starts: the tree build decides the waves' force precision with it."""
        self.sim.owner_set_dt(dt)

    def op_maxabs(self):
        self.sim.owner_maxabs(self.maxabs.data_ptr())

    def op_sample(self):
        # samples in proportion to the bodies held (a rank at 1.25 x its share fills every slot): equal quantiles of
        # the pooled samples are then equal shares of the BODIES, and an imbalance is corrected in one step
        share = max(1.0, self.n_total / self.world)
        nvalid = int(round(0.8 * self.SAMPLES * self.sim.n / share))
        self.sim.owner_sample(self.maxabs.data_ptr(), self.samples.data_ptr(), self.SAMPLES, max(1, min(self.SAMPLES, nvalid)))

    def op_partition(self, all_samples):
        return self.sim.owner_partition(all_samples.data_ptr(), all_samples.numel(), self.send_rows.data_ptr())

    def op_adopt(self, rows, n_recv):
        """`n_recv` immigrant rows join the bodies that stayed; keys, sort, octree, boxes."""
        self.sim.owner_adopt(rows.data_ptr(), n_recv, self.maxabs.data_ptr(), self.bbox.data_ptr(), self.chain.data_ptr())

    def op_export_let(self):
        return self.sim.owner_export_let(self.boxes.data_ptr(), self.chains.data_ptr(), self.let_send.data_ptr())

    def step_facts(self):
        """(asking waves, waves, tree rows that fit in front of / behind the own piece): see nbmi_owner_step_facts."""
        return self.sim.owner_step_facts()

    def op_step(self, recv_counts, dt, all64=None):
        self.sim.owner_set_all64(all64)
        self.sim.owner_step(self.let_recv.data_ptr(), recv_counts, dt)

    def wait(self):
        """A collective issued on torch's stream has finished (only needed while the library works on a stream of its
        own; with the shared stream the order is the stream's)."""
        if self.stream is None:
            self.torch.cuda.current_stream(self.device).synchronize()

    def stream_scope(self):
        """Context in which torch's current stream is the library's."""
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def owned_state(self):
        """(global ids, positions f64, velocities f64) of the owned bodies."""
        return self.sim.ids().astype(np.int64), self.sim.get_positions_f64(), self.sim.get_velocities()


class LetBarnesHut:
    """step() over `world` ranks in owner mode; see the module docstring.  The engine owns the buffers
    (maxabs, samples / all_samples, send_rows / recv_rows, bbox / boxes, let_send / let_recv: torch tensors)
    and the six phases op_*; `comm` provides all_reduce_max(t), all_gather(full, mine),
    all_to_all_counts(np int64[world]) -> np int64[world] and
    all_to_all_rows(recv, send, recv_counts, send_counts) (default: torch.distributed, DistComm)."""

    def __init__(self, engine, rank, world, comm=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.comm = comm
        self.n = engine.n_total
        self.all64 = False  # force precision "auto": the system-wide "every wave float64" state (hysteresis, see _verdict)
        self._carried = None  # a failure of the last op_step, to be raised by every rank at the next exchange
        if world > 1 and comm is None:
            raise ValueError("LetBarnesHut over more than one rank needs a communicator")

    def step(self, dt, substeps=1):
        scope = self.engine.stream_scope() if hasattr(self.engine, "stream_scope") else __import__("contextlib").nullcontext()
        with scope:
            self._step(dt, substeps)

    def _exchange_counts(self, send_counts, extras):
        """recv_counts for this rank, the whole (source x destination) matrix and every rank's `extras` (its row budget
        and the like) - via comm.counts_matrix when the communicator has it; a plain all-to-all of the counts otherwise."""
        W = self.world
        if hasattr(self.comm, "counts_matrix"):
            row = np.concatenate([np.asarray(send_counts, dtype=np.int64), np.asarray(extras, dtype=np.int64)])
            M = self.comm.counts_matrix(row)
            return M[:, self.rank].copy(), M[:, :W], M[:, W:]
        return self.comm.all_to_all_counts(send_counts), None, None

    def _raise_together(self, failed, extras, col, phase):
        """`failed`: this rank's own exception of the phase (or None); extras[:, col]: every rank's failure flag."""
        if extras is None:
            if failed is not None:
                raise failed
            return
        bad = np.nonzero(extras[:, col])[0]
        if len(bad):
            if isinstance(phase, dict):  # the flag's value says where that rank failed
                phase = phase.get(int(extras[bad[0], col]), "?")
            msg = f"owner mode: rank(s) {bad.tolist()} failed in the {phase} phase (every rank raises this together)"
            if failed is not None:
                raise RuntimeError(f"{msg}: {failed}") from failed
            raise RuntimeError(msg)

    def _verdict(self, ask, waves):
        """The single handle's rule on the ranks' summed votes (nbmi.hip all64_rule): every wave computes in float64
        from the step in which more than a third of the system's waves ask for it until fewer than a quarter do."""
        enter = float(os.environ.get("NBMI_ALL64_ENTER", ALL64_ENTER))
        leave = float(os.environ.get("NBMI_ALL64_LEAVE", ALL64_LEAVE))
        self.all64 = (ask >= leave * waves) if self.all64 else (ask > enter * waves)
        return self.all64

    def _step(self, dt, substeps):
        e, W = self.engine, self.world
        for _ in range(substeps):
            wire = 0
            if hasattr(e, "op_begin"):
                e.op_begin(dt)
            e.op_maxabs()
            if W > 1:
                self.comm.all_reduce_max(e.maxabs)
                e.wait()
                wire += 8
            e.op_sample()
            allsamp = e.samples
            if W > 1:
                self.comm.all_gather(e.all_samples, e.samples)
                e.wait()
                wire += 8 * e.SAMPLES
                allsamp = e.all_samples
            # A failure inside one rank's library call (the tree of the last step overflowed, a sort timed out ...) is
            # carried through the next exchange of counts as a flag: every rank raises, none is left in a collective
            failed, self._carried = self._carried, None
            where = 2 if failed is not None else 0
            try:
                send_counts = e.op_partition(allsamp)  # rows of the bodies that leave, grouped by destination
                if failed is not None:
                    send_counts = np.zeros(W, dtype=np.int64)
            except RuntimeError as exc:
                if W == 1:
                    raise
                failed, send_counts, where = exc, np.zeros(W, dtype=np.int64), 1
            n_recv = 0
            if W > 1:
                held = e.sim.n if hasattr(e, "sim") else 0  # (a stand-in engine has no row budget: 0 rows of "infinity")
                room = e.cap if hasattr(e, "sim") else (1 << 62)
                recv_counts, M, ex = self._exchange_counts(send_counts, [held, room, where])
                self._raise_together(failed, ex, 2, {1: "partition", 2: "walk (of the previous step)"})
                if M is not None:
                    # every rank checks EVERY rank's body rows against THAT rank's budget and raises with it
                    after = ex[:, 0] + M.sum(axis=0)  # rows a rank holds while it adopts: stayers, leavers' rows, arrivals
                    if (after > ex[:, 1]).any():
                        j = int(np.argmax(after - ex[:, 1]))
                        raise RuntimeError(f"owner mode: rank {j} would hold {int(after[j])} body rows, capacity {int(ex[j, 1])} "
                                           "(every rank raises this together; raise the head room in let_capacities)")
                n_recv = int(recv_counts.sum())
                self.comm.all_to_all_rows(e.recv_rows, e.send_rows, recv_counts, send_counts)
                e.wait()
                wire += int(send_counts.sum()) * ROW * 8
            rows, n_new = e.recv_rows, n_recv
            try:
                e.op_adopt(rows, n_new)
            except RuntimeError as exc:  # (travels with the tree counts below: every rank raises)
                if W == 1:
                    raise
                failed, where = exc, 2
            counts = np.zeros(W, dtype=np.int64)
            all64 = None
            if W > 1:
                self.comm.all_gather(e.boxes, e.bbox)
                if hasattr(e, "chain"):
                    self.comm.all_gather(e.chains, e.chain)
                    wire += e.chain.numel() * 8
                e.wait()
                let_counts = np.zeros(W, dtype=np.int64)
                facts = np.array([0, 0, 1 << 62, 1 << 62], dtype=np.int64)
                if failed is None:
                    try:
                        let_counts = e.op_export_let()  # rows for every other rank: only what THAT rank's bodies can open
                        if hasattr(e, "step_facts"):
                            facts = np.asarray(e.step_facts(), dtype=np.int64)
                    except RuntimeError as exc:
                        failed, let_counts, where = exc, np.zeros(W, dtype=np.int64), 1
                room = e.let_recv.shape[0]
                counts, M, ex = self._exchange_counts(let_counts, [room, where if failed else 0, *facts.tolist()])
                self._raise_together(failed, ex, 1, {1: "tree export", 2: "adopt"})
                if M is not None:  # the same verdict on every rank
                    incoming = M.sum(axis=0)
                    if (incoming > ex[:, 0]).any():
                        j = int(np.argmax(incoming - ex[:, 0]))
                        raise RuntimeError(f"owner mode: rank {j} would receive {int(incoming[j])} tree rows, {int(ex[j, 0])} "
                                           "reserved (every rank raises this together)")
                    # ... and whether the pieces fit around every rank's own tree (nbmi_owner_step's own check, which
                    # would fire on that rank alone): rows from lower ranks go in front of it, from higher ranks behind
                    lower = np.array([M[:j, j].sum() for j in range(W)])
                    upper = np.array([M[j + 1:, j].sum() for j in range(W)])
                    if (lower > ex[:, 4]).any() or (upper > ex[:, 5]).any():
                        j = int(np.argmax(np.maximum(lower - ex[:, 4], upper - ex[:, 5])))
                        raise RuntimeError(f"owner mode: the trees rank {j} receives do not fit around its own ({int(lower[j])} rows "
                                           f"in front, room {int(ex[j, 4])}; {int(upper[j])} behind, room {int(ex[j, 5])}; every "
                                           "rank raises this together)")
                    # force precision "auto": ONE decision for the system from the ranks' summed votes [r4] - a rank
                    # that applied the rule to its own waves made the arithmetic depend on the world size
                    if int(ex[:, 3].sum()) > 0:
                        all64 = self._verdict(int(ex[:, 2].sum()), int(ex[:, 3].sum()))
                elif int(counts.sum()) > room:
                    raise RuntimeError(f"rank {self.rank}: {int(counts.sum())} received tree rows exceed the {room} reserved")
                self.comm.all_to_all_rows(e.let_recv, e.let_send, counts, let_counts)
                e.wait()
                wire += e.bbox.numel() * 8 + int(let_counts.sum()) * e.LET_ROW_BYTES
            try:
                if all64 is None:
                    e.op_step(counts, dt)
                else:
                    e.op_step(counts, dt, all64=all64)
            except RuntimeError as exc:
                if W == 1:
                    raise
                self._carried = exc  # (what can fail on one rank alone now: a device error; raised by all next time)
            e.let_counts = counts
            e.wire_bytes = wire
            e.migrated = int(send_counts.sum())

    def gather_state(self):
        """Full (positions, velocities) float64 in the caller's original order, on every rank.  (A failure of this
        rank's last walk that no later step has announced to the others yet is raised here, on this rank: its state is
        not the state after that step.)"""
        if self._carried is not None:
            failed, self._carried = self._carried, None
            raise failed
        ids, pos, vel = self.engine.owned_state()
        if self.world == 1:
            out_p, out_v = np.empty((self.n, 3)), np.empty((self.n, 3))
            out_p[ids], out_v[ids] = pos, vel
            return out_p, out_v
        import torch
        cap = self.engine.cap
        mine = torch.full((cap, 7), -1.0, dtype=torch.float64)
        mine[: len(ids), 0] = torch.from_numpy(ids.astype(np.float64))
        mine[: len(ids), 1:4] = torch.from_numpy(pos)
        mine[: len(ids), 4:7] = torch.from_numpy(vel)
        dev = self.engine.maxabs.device
        mine = mine.to(dev)
        full = torch.empty((cap * self.world, 7), dtype=torch.float64, device=dev)
        self.comm.all_gather(full, mine)
        self.engine.wait()
        rows = full.cpu().numpy()
        rows = rows[rows[:, 0] >= 0]
        out_p, out_v = np.empty((self.n, 3)), np.empty((self.n, 3))
        gid = rows[:, 0].astype(np.int64)
        assert len(np.unique(gid)) == self.n, "every body must have exactly one owner"
        out_p[gid], out_v[gid] = rows[:, 1:4], rows[:, 4:7]
        return out_p, out_v


class DistComm:
    """The collectives of the owner mode on torch.distributed.  Backend "nccl" (= RCCL over xGMI) works on the
    device tensors directly; any other backend (gloo: CPU tests, 1-GPU rehearsals) goes through host copies."""

    def __init__(self, dist, device=None):
        self.dist = dist
        self.device = device
        self.direct = dist.get_backend() == "nccl"

    def _run(self, fn, outs, ins):
        """fn(*outs, *ins) on tensors the backend can take; results copied back into `outs`."""
        if self.direct or all(t.device.type == "cpu" for t in outs + ins):
            return fn(*outs, *ins)
        h_out = [t.cpu() for t in outs]
        fn(*h_out, *[t.cpu() for t in ins])
        for t, h in zip(outs, h_out):
            t.copy_(h)

    def all_reduce_max(self, t):
        self._run(lambda x: self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX), [t], [])

    def all_gather(self, full, mine):
        self._run(lambda f, m: self.dist.all_gather_into_tensor(f, m), [full], [mine])

    def _small(self, values):
        import torch
        dev = self.device if self.direct else "cpu"
        return torch.tensor(np.asarray(values, dtype=np.int64), dtype=torch.int64, device=dev)

    def all_to_all_counts(self, send_counts):
        import torch
        s = self._small(send_counts)
        r = torch.empty_like(s)
        self.dist.all_to_all_single(r, s)
        return r.cpu().numpy()

    def counts_matrix(self, send_row):
        """(world, len(send_row)) int64: every rank's row, on every rank.  With the whole matrix each rank can make the
        SAME decision about every other rank's buffers (a capacity error raised on one rank only leaves the
        others waiting in the next collective until its time-out)."""
        import torch
        s = self._small(send_row)
        r = torch.empty(self.dist.get_world_size() * s.numel(), dtype=torch.int64, device=s.device)
        self.dist.all_gather_into_tensor(r, s)
        return r.cpu().numpy().reshape(self.dist.get_world_size(), -1)

    def all_gather_counts(self, mine):
        import torch
        s = self._small([mine])
        r = torch.empty(self.dist.get_world_size(), dtype=torch.int64, device=s.device)
        self.dist.all_gather_into_tensor(r, s)
        return r.cpu().numpy()

    def all_to_all_rows(self, recv, send, recv_counts, send_counts):
        n_in, n_out = int(recv_counts.sum()), int(send_counts.sum())
        rs, ss = [int(c) for c in recv_counts], [int(c) for c in send_counts]
        if self.direct or recv.device.type == "cpu":
            self.dist.all_to_all_single(recv[:n_in], send[:n_out], output_split_sizes=rs, input_split_sizes=ss)
            return
        h = recv[:n_in].cpu()
        self.dist.all_to_all_single(h, send[:n_out].cpu(), output_split_sizes=rs, input_split_sizes=ss)
        recv[:n_in].copy_(h)


def unpack_rows(rows: np.ndarray):
    """(positions, velocities, masses, ids) from packed rows."""
    return rows[:, 0:3], rows[:, 3:6], rows[:, 6], rows[:, 7].astype(np.int64)
