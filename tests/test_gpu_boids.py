"""GPU parity tests for the boids neighbour sweep (bdmi_* C ABI / boids.Flock).

Tolerances: cell indices bit-exact (integer work); forces / state float64 with a different
within-cell summation order than np.argsort gives -> |diff| <= 1e-9 absolute (values O(1..500)).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


class RawFlock:
    """Thin driver of the C ABI with explicit state and params (golden cases override bounds)."""

    def __init__(self, nat, pos, vel, col, params):
        self.lib = nat.load()
        self.nat = nat
        self.n = len(pos)
        self.h = self.lib.bdmi_create(self.n, nat.ptr(pos), nat.ptr(vel), nat.ptr(col), nat.ptr(params), 0)
        assert self.h, nat.last_error()

    def step(self, dt, k=1):
        self.nat.check(self.lib.bdmi_step(self.h, dt, k), "bdmi_step")

    def state(self):
        out = [np.empty((self.n, 3)) for _ in range(3)]
        self.nat.check(self.lib.bdmi_get_state(self.h, *[self.nat.ptr(a) for a in out]), "bdmi_get_state")
        return out

    def cells(self):
        out = np.empty(self.n, dtype=np.int32)
        self.nat.check(self.lib.bdmi_get_cell_indices(self.h, self.nat.ptr(out)), "cells")
        return out

    def forces(self):
        out = [np.empty((self.n, 3)) for _ in range(4)]
        self.nat.check(self.lib.bdmi_get_forces(self.h, *[self.nat.ptr(a) for a in out]), "forces")
        return out

    def close(self):
        self.lib.bdmi_destroy(self.h)


@pytest.mark.parametrize("tag", ["sparse", "dense", "walls"])
def test_flock_vs_reference(gpu, oracle, tag):
    g = golden("boids_" + tag)
    params = oracle.boids_params(bounds=float(g["bounds"]))
    f = RawFlock(gpu, g["pos_0"], g["vel_0"], g["col_0"], params)
    assert np.array_equal(f.cells(), g["cell_indices_1"])  # assign_cells of the initial positions
    sep, ali, coh, avg = f.forces()
    for mine, key in ((sep, "sep_1"), (ali, "ali_1"), (coh, "coh_1"), (avg, "avg_1")):
        d = np.abs(mine - g[key]).max()
        print(tag, key, "max abs diff", d)
        assert d <= 1e-9
    dt = float(g["dt"])
    steps = int(g["steps"])
    f.step(dt)
    p, v, c = f.state()
    assert np.abs(p - g["pos_1"]).max() <= 1e-9 and np.abs(v - g["vel_1"]).max() <= 1e-9
    assert np.abs(c - g["col_1"]).max() <= 1e-12
    f.step(dt, steps - 1)
    p, v, c = f.state()
    dp, dv, dc = (np.abs(a - g[k + f"_{steps}"]).max() for a, k in ((p, "pos"), (v, "vel"), (c, "col")))
    print(tag, f"after {steps} steps: max abs diff pos {dp:.2e} vel {dv:.2e} col {dc:.2e}")
    assert dp <= 1e-8 and dv <= 1e-7 and dc <= 1e-10
    f.close()


def test_flock_class_api(gpu, oracle):
    import config.boids as bcfg
    from boids import Flock
    saved = dict(bcfg.BOIDS)
    bcfg.BOIDS["bounds"] = 30.0
    try:
        fl = Flock(8192, seed=7)
    finally:
        bcfg.BOIDS.clear()
        bcfg.BOIDS.update(saved)
    assert fl.positions.shape == (8192, 3) and fl.positions.dtype == np.float64
    assert fl.grid_dim == int(np.ceil(60 / 5.0)) + 2
    st = oracle.FlockStepper(fl.positions, fl.velocities, fl.colors, oracle.boids_params(bounds=30.0))
    for _ in range(5):
        fl.update(1.0 / 60.0)
        st.step(1.0 / 60.0)
    assert np.abs(fl.positions - st.pos).max() <= 1e-9
    assert np.abs(fl.velocities - st.vel).max() <= 1e-8
    assert np.abs(fl.colors - st.col).max() <= 1e-11
    info = fl.grid_info()
    assert info["num_cells"] == fl.num_cells and info["occupied"] == len(np.unique(st.cell_indices))
    fl.set_state(positions=st.pos * 0.5)
    assert np.array_equal(fl.positions, st.pos * 0.5) and np.array_equal(fl.velocities, fl.velocities)
    with pytest.raises(NotImplementedError):
        fl.draw()
    fl.close()


def test_edge_empty_single_and_clamped(gpu, oracle):
    params = oracle.boids_params(bounds=10.0)
    z = np.zeros((0, 3))
    f = RawFlock(gpu, z, z, z, params)
    f.step(0.01)
    f.close()
    one = RawFlock(gpu, np.array([[1.0, 2.0, 3.0]]), np.array([[1.0, 0.0, 0.0]]), np.array([[0.2, 0.4, 0.6]]), params)
    one.step(0.5)
    p, v, c = one.state()
    assert np.allclose(p, [[1.5, 2.0, 3.0]]) and np.allclose(c, [[0.2, 0.4, 0.6]])
    one.close()
    # far outside the grid: cells clamp to the border (flock.py:23-25), walls push back
    rng = np.random.RandomState(2)
    pos = rng.uniform(-40, 40, (512, 3))
    vel = rng.uniform(-5, 5, (512, 3))
    col = rng.uniform(0, 1, (512, 3))
    f = RawFlock(gpu, pos, vel, col, params)
    st = oracle.FlockStepper(pos, vel, col, params)
    st.L.bdref_assign_cells(st.pos, st.cell_indices, st.cell, st.dim, st.offset, st.n)
    assert np.array_equal(f.cells(), st.cell_indices)
    for _ in range(3):
        f.step(0.02)
        st.step(0.02)
    p, v, c = f.state()
    assert np.abs(p - st.pos).max() <= 1e-9 and np.abs(v - st.vel).max() <= 1e-9
    f.close()


def test_config5_two_million_boids_properties(gpu, oracle):
    """BASELINE config 5 size: 2 M boids, reference constants, dt = 1/60.  Size-independent
    properties + oracle comparison of one full step."""
    from boids import Flock
    fl = Flock(2_000_000, seed=42)
    assert fl.grid_dim == 202 and fl.num_cells == 8_242_408
    p0, v0, c0 = fl.positions.copy(), fl.velocities.copy(), fl.colors.copy()
    st = oracle.FlockStepper(p0, v0, c0, oracle.boids_params(), use_numpy_argsort=False)
    st.L.bdref_assign_cells(st.pos, st.cell_indices, st.cell, st.dim, st.offset, st.n)
    assert np.array_equal(fl.cell_indices(), st.cell_indices)  # bit-exact integer work at full size
    fl.update(1.0 / 60.0)
    st.step(1.0 / 60.0)
    assert np.abs(fl.positions - st.pos).max() <= 1e-9 and np.abs(fl.velocities - st.vel).max() <= 1e-8
    fl.update(1.0 / 60.0, substeps=5)
    sp = np.linalg.norm(fl.velocities, axis=1)
    assert np.isfinite(fl.positions).all() and sp.max() <= 25.0 * (1 + 1e-12)
    assert fl.colors.min() >= 0.0 and fl.colors.max() <= 1.0
    fl.close()


def test_slab_sharded_flock_equals_single_handle(gpu):
    """boids in x-slabs with a one-cell halo.  Three threads on one GPU play three ranks through
    SlabFlock.step (neighbour exchange = all-to-all-v between threads); after 25 steps every boid has exactly one
    owner, boids have migrated between slabs, and the state equals the single handle's to float64 summation
    order (same candidate sets)."""
    import threading
    import nbmi_native as nat
    from boids.flock import generate_initial_state
    from boids.sharded import HipSlabEngine, SlabFlock
    from test_gpu_sharded_record import _ThreadComm
    np.random.seed(3)
    n, bounds = 30_000, 60.0
    pos, vel, col = generate_initial_state(n, bounds, 25.0)
    from oracle import pyref
    params = pyref.boids_params(bounds=bounds)
    dt, steps, world = 1.0 / 60.0, 25, 3
    lib = nat.load()
    h = lib.bdmi_create(n, nat.ptr(pos), nat.ptr(vel), nat.ptr(col), nat.ptr(params), 0)
    nat.check(lib.bdmi_step(h, dt, steps), "bdmi_step")
    ref = [np.empty((n, 3)) for _ in range(3)]
    nat.check(lib.bdmi_get_state(h, nat.ptr(ref[0]), nat.ptr(ref[1]), nat.ptr(ref[2])), "bdmi_get_state")
    lib.bdmi_destroy(h)

    comm = _ThreadComm(world)
    engines = [HipSlabEngine(pos, vel, col, params, r, world) for r in range(world)]
    start_ids = [set(e.owned_rows()[:, 9].astype(int)) for e in engines]
    flocks = [SlabFlock(e, r, world, comm.bind(r)) for r, e in enumerate(engines)]
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            flocks[r].step(dt, steps)
            out[r] = flocks[r].gather_state(n)
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)
            comm.bar.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    end_ids = [set(e.owned_rows()[:, 9].astype(int)) for e in engines]
    moved = sum(len(a - b) for a, b in zip(start_ids, end_ids))
    print(f"slabs: owned {[len(s) for s in end_ids]}, boids that changed owner {moved}, halo rows sent by rank 1 last step "
          f"{engines[1].sent_rows}")
    assert moved > 0 and sum(len(s) for s in end_ids) == n
    for r in range(world):
        for got, want, what in zip(out[r], ref, ("positions", "velocities", "colors")):
            err = np.abs(got - want).max()
            assert err <= 1e-9, (r, what, err)
    for e in engines:
        e.close()
