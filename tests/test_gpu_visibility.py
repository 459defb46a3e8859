"""GPU parity of the render-side reductions: frustum culling + compaction on the
device against the reference's own outputs (tests/golden/visibility_*.npz, produced by
compute_visibility_points / compute_visibility_numba / build_vertices_numba) and against the oracle."""
import math
import time

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _cams(g):
    for k in range(4):
        cam = g[f"cam_{k}"]
        yield k, cam[0:3], cam[3:6], cam[6:9], cam[9:12], float(g[f"tan_{k}"][0]), float(g[f"tan_{k}"][1])


def test_nbody_visible_points_equal_reference_masks(gpu, oracle):
    from nbody.gpu_backend import HIPBarnesHutSimulation
    g = golden("visibility_nbody")
    pos = g["pos"]
    rng = np.random.RandomState(1)
    vel = rng.normal(0, 6, pos.shape)
    sim = HIPBarnesHutSimulation(pos, vel, np.ones(len(pos)), 0.15, 3.0, 1.0, 0.5)
    sim.compute_colors(15.0)
    col = sim.get_colors()
    for k, cp, cf, cr, cu, th, tv in _cams(g):
        p, c = sim.visible_points(cp, cf, cr, cu, th, tv, float(g["far"]))
        mask = g[f"mask_{k}"]
        assert len(p) == mask.sum()
        assert np.array_equal(p, pos[mask].astype(np.float32))  # body order kept, one rounding
        assert np.array_equal(c, col[mask])
    # after stepping: state lives in key order on the device, the output order must still be the caller's
    sim.step_many(0.05, 3)
    sim.compute_colors(15.0)
    p64, col = sim.get_positions_f64(), sim.get_colors()
    for k, cp, cf, cr, cu, th, tv in _cams(g):
        mask = oracle.compute_visibility_points(p64, cp, cf, cr, cu, th, tv, 5000.0)
        p, c = sim.visible_points(cp, cf, cr, cu, th, tv, 5000.0)
        assert np.array_equal(p, p64[mask].astype(np.float32)) and np.array_equal(c, col[mask])
    # nothing visible / everything visible
    p, c = sim.visible_points((0, 0, 1e7), (0, 0, 1), (1, 0, 0), (0, 1, 0), 1.0, 1.0, 5000.0)
    assert len(p) == 0 and len(c) == 0
    p, c = sim.visible_points((0, 0, -4000.0), (0, 0, 1), (1, 0, 0), (0, 1, 0), 10.0, 10.0, 1e9)
    assert len(p) == len(pos)


def test_nbody_simulation_class_visibility(gpu, oracle):
    from nbody.simulation import NBodySimulation
    sim = NBodySimulation(20_000, seed=3)
    sim.update(0.02)
    eye = np.array([0.0, 150.0, 700.0])
    f = -eye / np.linalg.norm(eye)
    r = np.cross(f, [0.0, 1.0, 0.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    vp, vc = sim.visible_arrays(eye, f, r, u, fov=60, aspect=16 / 9)
    hv = math.radians(60) / 2
    mask = oracle.compute_visibility_points(sim._gpu_sim.get_positions_f64(), eye, f, r, u,
                                            math.tan(math.atan(math.tan(hv) * 16 / 9)), math.tan(hv),
                                            sim.fog_end)
    assert sim._visible_count == mask.sum() == len(vp) and 0 < len(vp) < sim.num_bodies
    assert np.array_equal(vc, sim.colors[mask])
    allp, allc = sim.visible_arrays()
    assert len(allp) == sim.num_bodies and sim._visible_count == sim.num_bodies


def test_boids_visible_vertices_equal_reference(gpu, oracle):
    from boids.flock import Flock
    g = golden("visibility_boids")
    fl = Flock(len(g["pos"]), seed=1)
    fl.set_state(g["pos"], g["vel"], g["col"])
    assert float(fl.cone_length) == float(g["cone_length"]) and float(fl.cone_radius) == float(g["cone_radius"])
    lib, h = fl._lib, fl._h
    import ctypes as C
    import nbmi_native as nat
    n = fl.num_boids
    verts = np.zeros((6 * n, 3), np.float32)
    cols = np.zeros((6 * n, 3), np.float32)
    for k, cp, cf, cr, cu, th, tv in _cams(g):
        cam = np.ascontiguousarray(np.concatenate([cp, cf, cr, cu]))
        cnt = C.c_int64(0)
        nat.check(lib.bdmi_visible_vertices(h, nat.ptr(cam), th, tv, float(g["fog_end"]), float(g["cone_length"]),
                                            float(g["cone_radius"]), nat.ptr(verts), nat.ptr(cols), n,
                                            C.addressof(cnt)), "bdmi_visible_vertices")
        nv = 6 * int(cnt.value)
        assert int(cnt.value) == g[f"mask_{k}"].sum()
        assert np.array_equal(verts[:nv], g[f"vertices_{k}"])      # float32, bit for bit
        assert np.array_equal(cols[:nv], g[f"vert_colors_{k}"])
    # the class method derives the tangents like Flock._compute_visibility (fov_margin 1.15)
    fl.update(1 / 60, 3)  # boids now stored in cell order
    pos, vel, col = fl.positions, fl.velocities, fl.colors
    eye = np.array([30.0, 10.0, 45.0])
    f = -eye / np.linalg.norm(eye)
    r = np.cross(f, [0.0, 1.0, 0.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    v, c = fl.visible_vertices(eye, f, r, u, fov=75, aspect=16 / 9)
    hv = (math.radians(75) / 2) * 1.15
    mask = oracle.compute_visibility_boids(pos, eye, f, r, u, math.tan(math.atan(math.tan(hv) * 16 / 9)),
                                           math.tan(hv), fl.fog_end)
    ev, ec = oracle.build_vertices(pos, vel, col, np.where(mask)[0].astype(np.int32), float(fl.cone_length),
                                   float(fl.cone_radius))
    assert fl._visible_count == mask.sum() and 0 < mask.sum() < n
    assert np.array_equal(v, ev) and np.array_equal(c, ec)
    v, c = fl.visible_vertices()
    assert fl._visible_count == n and len(v) == 6 * n


def test_visibility_one_million_bodies_and_timing(gpu, oracle):
    """Full-size check (1 M bodies): exact mask equality against the oracle, plus the point of the
    row: time and bytes against fetching everything."""
    from nbody.gpu_backend import HIPBarnesHutSimulation
    from tools.presets import generate_distribution
    np.random.seed(42)
    p, v, m = generate_distribution("galaxy", 1_000_000, 800.0, 0.07)
    sim = HIPBarnesHutSimulation(p, v, m, 0.07, 1.5, 1.0, 0.5)
    sim.step_many(0.05, 2)
    sim.compute_colors(15.0)
    eye = np.array([100.0, 60.0, 250.0])
    f = -eye / np.linalg.norm(eye)
    r = np.cross(f, [0.0, 1.0, 0.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    th, tv = math.tan(math.atan(math.tan(math.radians(37.5)) * 16 / 9)), math.tan(math.radians(37.5))
    vp, vc = sim.visible_points(eye, f, r, u, th, tv, 5000.0)
    sim.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        vp, vc = sim.visible_points(eye, f, r, u, th, tv, 5000.0)
    t_dev = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    p64 = sim.get_positions_f64()
    col = sim.get_colors()
    t_fetch = time.perf_counter() - t0
    t0 = time.perf_counter()
    mask = oracle.compute_visibility_points(p64, eye, f, r, u, th, tv, 5000.0)
    ref_p, ref_c = p64[mask].astype(np.float32), col[mask]
    t_cpu = time.perf_counter() - t0
    assert np.array_equal(vp, ref_p) and np.array_equal(vc, ref_c)
    print(f"visible {mask.sum()} of 1M: device cull+compact+D2H {1e3 * t_dev:.2f} ms; "
          f"fetch-all {1e3 * t_fetch:.2f} ms + CPU cull/gather {1e3 * t_cpu:.2f} ms")


def test_visibility_edge_cases(gpu, oracle):
    """Empty handles, a capacity smaller than the visible count, bodies exactly on the planes."""
    import ctypes as C
    import nbmi_native as nat
    from boids.flock import Flock
    from nbody.gpu_backend import HIPBarnesHutSimulation
    z3 = np.zeros((0, 3))
    empty = HIPBarnesHutSimulation(z3, z3, np.zeros(0), 1.0, 1.0, 1.0, 0.5)
    p, c = empty.visible_points((0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0), 1.0, 1.0, 100.0)
    assert p.shape == (0, 3) and c.shape == (0, 3)
    # bodies on the near plane (z = 0.1 is visible: the test is z < 0.1), the far plane (z = far visible), the
    # side planes (|x| < half_width is strict) - float64 exact values
    pos = np.array([[0, 0, 0.1], [0, 0, 0.1], [0, 0, np.nextafter(0.1, 0)], [0, 0, 50.0], [0, 0, np.nextafter(50.0, 100)],
                    [1.2 * 2.0, 0, 2.0], [np.nextafter(2.4, 0), 0, 2.0], [0, -np.nextafter(2.4, 0), 2.0]], dtype=np.float64)
    sim = HIPBarnesHutSimulation(pos, np.zeros_like(pos), np.ones(len(pos)), 1.0, 1.0, 1.0, 0.5)
    sim.compute_colors(15.0)
    vp, _ = sim.visible_points((0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0), 1.0, 1.0, 50.0)
    expect = np.array([True, True, False, True, False, False, True, True])
    assert np.array_equal(expect, oracle.compute_visibility_points(pos, (0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0), 1.0, 1.0, 50.0))
    assert np.array_equal(vp, pos[expect].astype(np.float32))
    # capacity below the count: count is still reported, only `capacity` rows are written
    lib, h = sim._lib, sim._h
    cam = np.array([0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0], dtype=np.float64)
    outp, outc = np.full((2, 3), -7, np.float32), np.full((2, 3), -7, np.float32)
    cnt = C.c_int64(0)
    nat.check(lib.nbmi_visible_points(h, nat.ptr(cam), 1.0, 1.0, 50.0, nat.ptr(outp), nat.ptr(outc), 2, C.addressof(cnt)), "vis")
    assert cnt.value == expect.sum() and np.array_equal(outp, pos[expect][:2].astype(np.float32))
    # count only
    nat.check(lib.nbmi_visible_points(h, nat.ptr(cam), 1.0, 1.0, 50.0, None, None, 0, C.addressof(cnt)), "vis")
    assert cnt.value == expect.sum()
    fl = Flock(0, seed=1)
    v, c = fl.visible_vertices((0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 0), fov=75, aspect=1.5)
    assert len(v) == 0 and fl._visible_count == 0
