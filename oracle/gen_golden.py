#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE - runs only in the build container).

Imports the reference's own hot-path functions from /root/reference and dumps
small input/expected-output fixtures into tests/golden/.  The reference needs
`numba` and `PyOpenGL`, neither of which is installed (ordinary ImportError,
SURVEY.md section 8c), so a throw-away stub directory is created in a temp dir
at run time: `numba.njit` = identity decorator, `prange = range`.  The
reference functions then execute as plain CPython in strict IEEE-754 float64,
in source order.  Nothing of the reference (source, bytecode, stubs importing
it) is written into this repository - only data.

Functions exercised (reference file:line):
  nbody/simulation.py:63   build_octree
  nbody/simulation.py:201  compute_forces_barnes_hut
  nbody/simulation.py:281  update_positions_velocities
  nbody/simulation.py:308  compute_bounds
  nbody/simulation.py:320  compute_colors_by_velocity
  boids/flock.py:454       Flock (+ the five @njit kernels it drives)
  tools/presets.py:91      generate_distribution (galaxy / collision / cluster)
  tools/record.py:88       save_frame
  nbody/simulation.py:403  compute_visibility_points
  boids/flock.py:311,:351  compute_visibility_numba, build_vertices_numba

Usage:  python oracle/gen_golden.py [--only NAME ...] [--procs 8]
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def _install_stubs():
    d = tempfile.mkdtemp(prefix="nbmi_oracle_stubs_")
    os.makedirs(os.path.join(d, "numba"))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(
            "def njit(*a, **k):\n"
            "    if len(a) == 1 and callable(a[0]) and not k:\n"
            "        return a[0]\n"
            "    return lambda fn: fn\n"
            "prange = range\nint32 = float32 = None\n"
        )
    os.makedirs(os.path.join(d, "OpenGL"))
    open(os.path.join(d, "OpenGL", "__init__.py"), "w").close()
    with open(os.path.join(d, "OpenGL", "GL.py"), "w") as f:
        f.write("GL_DYNAMIC_DRAW = 0\n")
    with open(os.path.join(d, "OpenGL", "arrays.py"), "w") as f:
        f.write("vbo = None\n")
    os.makedirs(os.path.join(d, "zstandard"))
    open(os.path.join(d, "zstandard", "__init__.py"), "w").close()
    sys.dont_write_bytecode = True
    sys.path[:0] = [REF, d]


_install_stubs()
import nbody.simulation as refsim  # noqa: E402
import tools.presets as refpresets  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def alloc_nodes(n, rows=None):
    # as tools/record.py:795-802 (rows= override: the reference's 4N rows are too few for
    # deliberately deep edge-case inputs, where it would write out of bounds)
    m = rows or min(8_000_000, max(n * 4, 64))
    return dict(
        centers=np.zeros((m, 3)), half=np.zeros(m), mass=np.zeros(m), com=np.zeros((m, 3)),
        children=np.full((m, 8), -1, dtype=np.int32), body=np.full(m, -1, dtype=np.int32),
        leaf=np.ones(m, dtype=np.bool_),
    )


def ref_build(pos, mass, nd):
    n = len(pos)
    bounds = refsim.compute_bounds(pos, n)
    nd["children"].fill(-1)
    nd["body"].fill(-1)
    nd["leaf"].fill(True)
    nn = refsim.build_octree(pos, mass, n, bounds, nd["centers"], nd["half"], nd["mass"], nd["com"],
                             nd["children"], nd["body"], nd["leaf"])
    return bounds, nn


def tree_facts(nd, nn, n):
    """(level, path-key) of every node by walking node_children from the root."""
    level = np.zeros(nn, dtype=np.int32)
    key = [0] * nn
    order = [0]
    k = 0
    while k < len(order):
        u = order[k]
        k += 1
        for c in range(8):
            v = int(nd["children"][u, c])
            if v >= 0:
                level[v] = level[u] + 1
                key[v] = (key[u] << 3) | c
                order.append(v)
    assert len(order) == nn
    assert max(level) <= 21
    key = np.array(key, dtype=np.uint64)
    idx = np.lexsort((key, level))
    cells = np.stack([level[idx].astype(np.uint64), key[idx]], axis=1)
    leaf_level = np.full(n, -1, dtype=np.int32)
    leaf_key = np.zeros(n, dtype=np.uint64)
    for u in range(nn):
        b = int(nd["body"][u])
        if nd["leaf"][u] and b >= 0:
            leaf_level[b] = level[u]
            leaf_key[b] = key[u]
    return dict(
        num_nodes=np.int64(nn), max_depth=np.int32(level.max()),
        level_hist=np.bincount(level, minlength=24).astype(np.int64),
        cells_sha=sha(cells), cells=cells,
        leaf_level=leaf_level, leaf_key=leaf_key,
        root_mass=np.float64(nd["mass"][0]), root_com=nd["com"][0].copy(),
        n_internal=np.int64(int((~nd["leaf"][:nn]).sum())),
    )


# ---- parallel walk (the reference's prange loop split over processes) -------
_G = {}


def _walk_range(args):
    lo, hi = args
    g = _G
    n = hi - lo
    # the reference function indexes positions[i] for i in range(num_bodies): give it a
    # window by passing shifted views is not possible (leaf body_idx compare uses i), so
    # run it on the full arrays but only for this range via a thin loop wrapper:
    acc = np.zeros((g["n"], 3))
    _forces_range(g, acc, lo, hi)
    return lo, hi, acc[lo:hi]


def _forces_range(g, acc, lo, hi):
    """Call the reference walk for bodies [lo, hi) only.

    compute_forces_barnes_hut loops `for i in prange(num_bodies)`; under the stub
    `prange` is looked up as a module global of nbody.simulation at call time, so a
    range-restricting callable is installed for the duration of the call."""
    saved = refsim.prange
    refsim.prange = lambda nb: range(lo, hi)
    try:
        nd = g["nd"]
        refsim.compute_forces_barnes_hut(
            g["pos"], g["mass"], acc, nd["centers"], nd["half"], nd["mass"], nd["com"],
            nd["children"], nd["body"], nd["leaf"], g["nn"], g["n"], g["theta"], g["G"], g["eps"])
    finally:
        refsim.prange = saved


def ref_forces(pos, mass, nd, nn, theta, G, eps, procs):
    n = len(pos)
    acc = np.zeros((n, 3))
    if procs <= 1 or n < 512:
        refsim.compute_forces_barnes_hut(pos, mass, acc, nd["centers"], nd["half"], nd["mass"], nd["com"],
                                         nd["children"], nd["body"], nd["leaf"], nn, n, theta, G, eps)
        return acc
    _G.update(pos=pos, mass=mass, nd=nd, nn=nn, n=n, theta=theta, G=G, eps=eps)
    edges = np.linspace(0, n, procs * 4 + 1).astype(int)
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        for lo, hi, a in pool.imap_unordered(_walk_range, list(zip(edges[:-1], edges[1:]))):
            acc[lo:hi] = a
    return acc


def gen_ic(dist, n, R, G, seed=42):
    np.random.seed(seed)
    p, v, m = refpresets.generate_distribution(dist, n, R, G)
    return p.astype(np.float64), v.astype(np.float64), m.astype(np.float64)


def save(name, **kw):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB", flush=True)


# ---------------------------------------------------------------------------
def job_ic(args):
    """IC generator pins: hashes + head/tail rows (the build has its own generators)."""
    out = {}
    for dist, n, R, G in [("galaxy", 10_000, 500.0, 0.15), ("collision", 10_000, 2000.0, 0.08),
                          ("cluster", 4096, 300.0, 0.05), ("galaxy", 2048, 500.0, 0.15),
                          ("collision", 2048, 2000.0, 0.08), ("cluster", 2048, 300.0, 0.05),
                          ("galaxy", 256, 500.0, 0.15), ("galaxy", 100_000, 500.0, 0.15)]:
        p, v, m = gen_ic(dist, n, R, G)
        tag = f"{dist}_{n}"
        out[tag + "_pos_sha"] = sha(p)
        out[tag + "_vel_sha"] = sha(v)
        out[tag + "_mass_sha"] = sha(m)
        out[tag + "_pos_head"] = p[:64].copy()
        out[tag + "_pos_tail"] = p[-64:].copy()
        out[tag + "_vel_head"] = v[:64].copy()
        out[tag + "_vel_tail"] = v[-64:].copy()
        out[tag + "_params"] = np.array([R, G])
    save("ic_pins", **out)


def job_tree(args):
    """Per-IC tree facts + step-0 accelerations at theta 0.5 / 0.95 (N = 256, 2048)."""
    for dist, n, R, G, eps in [("galaxy", 256, 500.0, 0.15, 3.0), ("galaxy", 2048, 500.0, 0.15, 3.0),
                               ("collision", 2048, 2000.0, 0.08, 6.0), ("cluster", 2048, 300.0, 0.05, 1.0)]:
        p, v, m = gen_ic(dist, n, R, G)
        nd = alloc_nodes(n)
        bounds, nn = ref_build(p, m, nd)
        tf = tree_facts(nd, nn, n)
        a05 = ref_forces(p, m, nd, nn, 0.5, G, eps, args.procs)
        a095 = ref_forces(p, m, nd, nn, 0.95, G, eps, args.procs)
        # node arrays in the reference's own numbering (small N only) for a direct
        # field-by-field check of the serial-insertion restatement
        extra = {}
        if n <= 2048:
            extra = dict(node_centers=nd["centers"][:nn].copy(), node_half=nd["half"][:nn].copy(),
                         node_mass=nd["mass"][:nn].copy(), node_com=nd["com"][:nn].copy(),
                         node_children=nd["children"][:nn].copy(), node_body=nd["body"][:nn].copy(),
                         node_leaf=nd["leaf"][:nn].copy())
        save(f"tree_{dist}_{n}", pos=p, vel=v, mass=m, bounds=np.float64(bounds), G=np.float64(G),
             eps=np.float64(eps), acc_t050=a05, acc_t095=a095, **tf, **extra)


def job_edge(args):
    """Edge cases the reference handles: N=1, N=2, heavy masses, bodies on cell boundaries."""
    rng = np.random.RandomState(7)
    cases = {}
    cases["n1"] = (np.array([[1.0, -2.0, 3.0]]), np.array([2.5]))
    cases["n2"] = (np.array([[1.0, -2.0, 3.0], [-4.0, 0.5, 0.25]]), np.array([1.0, 3.0]))
    # lattice points exactly on octant planes of the root cube (x >= cx convention)
    g = np.array([-8.0, -4.0, 0.0, 4.0, 8.0])
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    cases["lattice"] = (lat, np.ones(len(lat)))
    # close pairs: deep subdivision (separation 1e-3 in a 100-wide box)
    base = rng.uniform(-50, 50, (64, 3))
    close = np.concatenate([base, base + rng.uniform(-1e-3, 1e-3, (64, 3))])
    cases["close_pairs"] = (close, rng.uniform(0.5, 2.0, 128))
    # live-mode galaxy masses: a few 100.0 heavies (nbody/simulation.py:582-585)
    pm = rng.normal(0, 30, (512, 3))
    mm = np.ones(512)
    mm[rng.choice(512, 5, replace=False)] = 100.0
    cases["heavy"] = (pm, mm)
    out = {}
    for tag, (p, m) in cases.items():
        p = np.ascontiguousarray(p, dtype=np.float64)
        m = np.ascontiguousarray(m, dtype=np.float64)
        n = len(p)
        nd = alloc_nodes(n, rows=8192)
        bounds, nn = ref_build(p, m, nd)
        tf = tree_facts(nd, nn, n)
        acc = ref_forces(p, m, nd, nn, 0.5, 1.0, 0.1, 1)
        out[tag + "_pos"] = p
        out[tag + "_mass"] = m
        out[tag + "_bounds"] = np.float64(bounds)
        out[tag + "_acc"] = acc
        for k in ("num_nodes", "max_depth", "level_hist", "cells", "leaf_level", "leaf_key", "root_mass", "root_com"):
            out[f"{tag}_{k}"] = tf[k]
    save("tree_edge_cases", **out)


def run_steps(p, v, m, theta, G, eps, damping, dt, steps, snaps, procs, label):
    n = len(p)
    p = p.copy()
    v = v.copy()
    nd = alloc_nodes(n)
    out = {}
    nn_hist = []
    t0 = time.time()
    for s in range(1, steps + 1):
        bounds, nn = ref_build(p, m, nd)
        nn_hist.append(nn)
        acc = ref_forces(p, m, nd, nn, theta, G, eps, procs)
        refsim.update_positions_velocities(p, v, acc, damping, dt, n)
        if s in snaps:
            col = np.zeros((n, 3), dtype=np.float32)
            refsim.compute_colors_by_velocity(v, col, n, 15.0)
            out[f"pos_{s}"] = p.copy()
            out[f"vel_{s}"] = v.copy()
            out[f"col_{s}"] = col
        if s % 10 == 0:
            print(f"    [{label}] step {s}/{steps}  {time.time() - t0:.0f}s", flush=True)
    out["num_nodes_per_step"] = np.array(nn_hist, dtype=np.int64)
    return out


def job_traj2048(args):
    """galaxy N=2048, config-1 constants, 100 steps (positions/velocities/colours at 1,10,100)."""
    p, v, m = gen_ic("galaxy", 2048, 500.0, 0.15)
    out = run_steps(p, v, m, 0.5, 0.15, 3.0, 1.0, 0.2, 100, {1, 10, 100}, args.procs, "galaxy2048")
    save("traj_galaxy_2048", pos_0=p, vel_0=v, mass=m, theta=0.5, G=0.15, eps=3.0, damping=1.0, dt=0.2, **out)


def job_traj10k(args):
    """BASELINE config 1: quick_galaxy, 10 K bodies, theta 0.5, dt 0.2, 100 steps."""
    p, v, m = gen_ic("galaxy", 10_000, 500.0, 0.15)
    out = run_steps(p, v, m, 0.5, 0.15, 3.0, 1.0, 0.2, 100, {1, 10, 100}, args.procs, "galaxy10k")
    # ICs are regenerated by the build's own generator (pinned in ic_pins); keep fixture small:
    keep = {k: (a.astype(np.float64) if k.startswith(("pos_100", "vel_100")) else a[::8])
            for k, a in out.items() if k != "num_nodes_per_step"}
    save("traj_galaxy_10k", num_nodes_per_step=out["num_nodes_per_step"], theta=0.5, G=0.15, eps=3.0,
         damping=1.0, dt=0.2, stride=8, **keep)


def job_tree10k(args):
    p, v, m = gen_ic("galaxy", 10_000, 500.0, 0.15)
    nd = alloc_nodes(len(p))
    bounds, nn = ref_build(p, m, nd)
    tf = tree_facts(nd, nn, len(p))
    acc = ref_forces(p, m, nd, nn, 0.5, 0.15, 3.0, args.procs)
    tf.pop("cells")
    save("tree_galaxy_10k", bounds=np.float64(bounds), acc_t050=acc, **tf)


def job_tree100k(args):
    """Tree facts only (no walk) for galaxy 100 K - pins node count / depth at a larger size."""
    p, v, m = gen_ic("galaxy", 100_000, 500.0, 0.15)
    nd = alloc_nodes(len(p))
    bounds, nn = ref_build(p, m, nd)
    tf = tree_facts(nd, nn, len(p))
    tf.pop("cells")
    leaf_level = tf.pop("leaf_level")
    leaf_key = tf.pop("leaf_key")
    # accelerations of a 512-body sample at theta=0.5
    acc = np.zeros((len(p), 3))
    sample = np.arange(0, len(p), len(p) // 512)[:512]
    g = dict(pos=p, mass=m, nd=nd, nn=nn, n=len(p), theta=0.5, G=0.15, eps=3.0)
    saved = refsim.prange
    refsim.prange = lambda nb: sample.tolist()
    try:
        refsim.compute_forces_barnes_hut(p, m, acc, nd["centers"], nd["half"], nd["mass"], nd["com"],
                                         nd["children"], nd["body"], nd["leaf"], nn, len(p), 0.5, 0.15, 3.0)
    finally:
        refsim.prange = saved
    save("tree_galaxy_100k", bounds=np.float64(bounds), sample=sample, acc_sample=acc[sample],
         leaf_level_sha=sha(leaf_level), leaf_key_sha=sha(leaf_key), **tf)


def job_colors(args):
    """Colour ramp known answers over the full t range incl. every break point."""
    t = np.concatenate([np.linspace(0, 1.2, 481), [0.15, 0.30, 0.45, 0.55, 0.90, 0.95, 0.99, 1.0]])
    vel = np.zeros((len(t), 3))
    vel[:, 0] = t * 15.0 * 0.6
    vel[:, 2] = t * 15.0 * 0.8
    col = np.zeros((len(t), 3), dtype=np.float32)
    refsim.compute_colors_by_velocity(vel, col, len(t), 15.0)
    save("colors_ramp", vel=vel, colors=col, max_speed=15.0)


def job_direct(args):
    """Direct-N^2 accelerations.  The reference kernels are Numba-CUDA closures
    (nbody/gpu_backend.py:145-240) and cannot execute here; their sum
    a_i = sum_{j!=i} G m_j d (|d|^2+eps^2)^(-3/2) is evaluated in float64 NumPy."""
    for dist, n, R, G, eps in [("cluster", 2048, 300.0, 0.05, 1.0), ("galaxy", 2048, 500.0, 0.15, 3.0)]:
        p, v, m = gen_ic(dist, n, R, G)
        d = p[None, :, :] - p[:, None, :]
        r2 = (d * d).sum(-1) + eps * eps
        inv = 1.0 / np.sqrt(r2)
        w = G * m[None, :] * inv * inv * inv
        np.fill_diagonal(w, 0.0)
        acc = (w[:, :, None] * d).sum(1)
        # one kick-drift step as update_bodies_cuda (gpu_backend.py:243-257)
        dt, damping = 0.02, 1.0
        v1 = (v + acc * dt) * damping
        p1 = p + v1 * dt
        save(f"direct_{dist}_{n}", pos=p, vel=v, mass=m, G=G, eps=eps, acc=acc, dt=dt, damping=damping,
             pos_1=p1, vel_1=v1)


def job_boids(args):
    """boids.Flock driven exactly as core/application.py does (update(dt) in a loop)."""
    import config.boids as bcfg
    import boids.flock as refflock
    for tag, n, bounds, steps in [("sparse", 4096, 40.0, 10), ("dense", 4096, 20.0, 10), ("walls", 1024, 8.0, 20)]:
        saved = dict(bcfg.BOIDS)
        bcfg.BOIDS["bounds"] = bounds
        try:
            np.random.seed(42)
            fl = refflock.Flock(n)
        finally:
            bcfg.BOIDS.clear()
            bcfg.BOIDS.update(saved)
        out = dict(pos_0=fl.positions.copy(), vel_0=fl.velocities.copy(), col_0=fl.colors.copy(),
                   bounds=bounds, dt=1.0 / 60.0, grid_dim=fl.grid_dim, cell_size=fl.cell_size,
                   grid_offset=fl.grid_offset)
        dt = 1.0 / 60.0
        for s in range(1, steps + 1):
            fl.update(dt)
            if s == 1:
                out.update(cell_indices_1=fl._cell_indices.copy(), cell_counts_1_sha=sha(fl._cell_counts),
                           sep_1=fl._sep_forces.copy(), ali_1=fl._align_forces.copy(),
                           coh_1=fl._coh_forces.copy(), avg_1=fl._avg_colors.copy(),
                           cell_counts_1_nonzero=np.flatnonzero(fl._cell_counts).astype(np.int64),
                           cell_counts_1_values=fl._cell_counts[np.flatnonzero(fl._cell_counts)].copy())
            if s in (1, steps):
                out.update({f"pos_{s}": fl.positions.copy(), f"vel_{s}": fl.velocities.copy(),
                            f"col_{s}": fl.colors.copy()})
        out["steps"] = steps
        save(f"boids_{tag}", **out)


def job_frame(args):
    """A raw frame written by the reference's own save_frame + delta-codec payload expectations."""
    import tools.record as refrec
    from pathlib import Path
    rng = np.random.RandomState(3)
    pos = rng.normal(0, 100, (16, 3))
    col = rng.uniform(0, 1, (16, 3)).astype(np.float32)
    d = Path(GOLD) / "frames"
    d.mkdir(exist_ok=True)
    refrec.save_frame(d, 0, pos, col)
    pos2 = pos + rng.normal(0, 0.5, (16, 3))
    pos2[3, 0] += 40.0  # forces the int16 wrap quirk (delta*1000 > 32767)
    col2 = np.clip(col + rng.normal(0, 0.01, (16, 3)), 0, 1).astype(np.float32)
    refrec.save_frame(d, 1, pos2, col2)
    p0, c0 = refrec.load_frame(d, 0)
    p1, c1 = refrec.load_frame(d, 1)
    # format-2 payload before zstd (tools/record.py:254-262) - plain NumPy in the reference
    with np.errstate(all="ignore"):
        dpos = ((p1 - p0) * 1000).astype(np.int16)
        dcol = ((c1 - c0) * 1000).astype(np.int16)
    save("frame_codec", pos64_0=pos, pos64_1=pos2, p0=p0, c0=c0, p1=p1, c1=c1, dpos_i16=dpos, dcol_i16=dcol,
         dec_p1=(p0 + dpos.astype(np.float32) / 1000.0), dec_c1=(c0 + dcol.astype(np.float32) / 1000.0))


def _cameras():
    """Four camera poses: outside looking at the origin, inside, grazing, narrow field."""
    cams = []
    for eye, target, fov_deg, aspect in [((0.0, 120.0, 600.0), (0.0, 0.0, 0.0), 75.0, 16 / 9),
                                         ((15.0, 3.0, -20.0), (100.0, 10.0, 40.0), 75.0, 16 / 9),
                                         ((-300.0, 2.0, 0.0), (300.0, 0.0, 5.0), 60.0, 4 / 3),
                                         ((40.0, 35.0, 40.0), (0.0, 0.0, 0.0), 20.0, 1.0)]:
        eye, target = np.array(eye), np.array(target)
        f = target - eye
        f /= np.linalg.norm(f)
        r = np.cross(f, np.array([0.0, 1.0, 0.0]))
        r /= np.linalg.norm(r)
        u = np.cross(r, f)
        cams.append((eye, f, r, u, np.radians(fov_deg), aspect))
    return cams


def job_visibility(args):
    """Render-side reductions: compute_visibility_points (nbody/simulation.py:403) and
    compute_visibility_numba + build_vertices_numba (boids/flock.py:311, :351), called with the
    tangents NBodySimulation._compute_visibility (:880-903) / Flock._compute_visibility (:680-709)
    derive from (fov, aspect)."""
    import math
    import config.boids as bcfg
    import boids.flock as refflock
    p, v, m = gen_ic("galaxy", 2048, 500.0, 0.15)
    out = dict(pos=p)
    for k, (eye, f, r, u, fov, aspect) in enumerate(_cameras()):
        half_v = fov / 2
        half_h = math.atan(math.tan(half_v) * aspect)
        mask = np.zeros(len(p), dtype=np.bool_)
        refsim.compute_visibility_points(p, eye, f, r, u, math.tan(half_h), math.tan(half_v), 5000.0, mask, len(p))
        out.update({f"cam_{k}": np.concatenate([eye, f, r, u]), f"tan_{k}": np.array([math.tan(half_h), math.tan(half_v)]),
                    f"mask_{k}": mask})
    out["far"] = 5000.0
    save("visibility_nbody", **out)

    saved = dict(bcfg.BOIDS)
    bcfg.BOIDS["bounds"] = 40.0
    try:
        np.random.seed(42)
        fl = refflock.Flock(4096)
    finally:
        bcfg.BOIDS.clear()
        bcfg.BOIDS.update(saved)
    fl.velocities[5] = 0.0           # speed floor branch
    fl.velocities[6] = (0.0, 7.0, 0.0)   # forward parallel to world-up: world-right branch
    out = dict(pos=fl.positions.copy(), vel=fl.velocities.copy(), col=fl.colors.copy(),
               cone_length=float(fl.cone_length), cone_radius=float(fl.cone_radius), fog_end=float(fl.fog_end))
    for k, (eye, f, r, u, fov, aspect) in enumerate(_cameras()):
        eye = eye * 0.1  # the flock lives in +-40
        fl._compute_visibility(eye, f, r, u, fov, aspect)
        nv = fl._build_vertices()
        half_v = (fov / 2) * fl.fov_margin
        half_h = math.atan(math.tan(half_v) * aspect)
        out.update({f"cam_{k}": np.concatenate([eye, f, r, u]), f"tan_{k}": np.array([math.tan(half_h), math.tan(half_v)]),
                    f"mask_{k}": fl._visible_mask.copy(), f"vertices_{k}": fl._vertices[:nv].copy(),
                    f"vert_colors_{k}": fl._vert_colors[:nv].copy()})
    save("visibility_boids", **out)


JOBS = dict(visibility=job_visibility, ic=job_ic, tree=job_tree, edge=job_edge, colors=job_colors, direct=job_direct, boids=job_boids,
            frame=job_frame, traj2048=job_traj2048, tree10k=job_tree10k, tree100k=job_tree100k,
            traj10k=job_traj10k)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in JOBS.items():
        if a.only and name not in a.only:
            continue
        t0 = time.time()
        print(f"[gen_golden] {name}", flush=True)
        fn(a)
        print(f"[gen_golden] {name} done in {time.time() - t0:.1f}s", flush=True)
    man_path = os.path.join(GOLD, "MANIFEST.json")
    keep = {}
    if os.path.exists(man_path):  # "oracle_cache": hashes of tests/cache/* (scripts/oracle_cache.py) - not ours to drop
        with open(man_path) as f:
            keep = {k: v for k, v in json.load(f).items() if k == "oracle_cache"}
    with open(man_path, "w") as f:
        json.dump({**{fn_: os.path.getsize(os.path.join(GOLD, fn_)) for fn_ in sorted(os.listdir(GOLD))
                      if fn_.endswith(".npz")}, **keep}, f, indent=1, sort_keys=True)
