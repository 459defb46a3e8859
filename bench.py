#!/usr/bin/env python3
"""Benchmark of the hot path: Barnes-Hut body-steps/sec at theta = 0.5 on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full timestep of the workload (bounds -> keys -> sort -> octree -> walk with
fused kick-drift) with the bodies already resident in HBM.  Default workload = BASELINE config 2
(galaxy, 1 M bodies per GPU, theta 0.5, dt 0.05, G 0.07, eps 1.5, R 800; synthetic IC from the
reference's generator restated in tools/presets.py, seed 42).  With N > 1 the work per GPU is
fixed (weak scaling): N x 1 M bodies, every rank owns one octant-key range (re-balanced every step, bodies
migrate by an all-to-all), builds the octree of its own bodies and receives the other ranks' locally
essential trees by an RCCL all-gather (nbody/sharded.py "let"; NBMI_SHARD_MODE=rows selects the
replicated-tree exchange that is bit-identical to one GPU).  `python bench.py --gpus N` typed plainly
starts its own N ranks.

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  roofline     - dominant kernel (k_walk): algorithmic bytes per launch / its mean duration,
                 measured with HIP events on the library's own stream (nbmi timers);
  cpu_baseline - the CPU oracle ("port" of the reference's CPU path, OpenMP, -O3 -ffast-math like
                 Numba fastmath) timed on this host for a bounded sample of the same workload.
"""
import argparse
import contextlib
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _self_launch():
    """`python bench.py --gpus N` typed plainly (no WORLD_SIZE in the environment, N > 1): start the N
    ranks as children through torch.distributed.run and hand back rank 0's JSON line.  This runs BEFORE
    torch or the HIP library is imported - a process that has touched the GPU must never be replaced or
    forked into ranks - and it starts children (no exec)."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 400))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402

sys.path.insert(0, ROOT)
importlib.import_module("3d-spatial-sim-for-boid-and-nbody_amd")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
NODE_BYTES = 24  # csrc/nbmi.hip Node
BODY_BYTES_WALK = 156  # walk kernel per body: reads posm 16 + perm 4 + state 60, writes 60 (+16 slack)

WORKLOADS = {
    # name: (distribution, bodies per GPU, R, G, eps, theta, dt, method)
    "galaxy_1m_bh": ("galaxy", 1_000_000, 800.0, 0.07, 1.5, 0.5, 0.05, "barnes_hut"),
    "collision_10m_bh": ("collision", 10_000_000, 2000.0, 0.08, 6.0, 0.5, 0.25, "barnes_hut"),
    "galaxy_10k_bh": ("galaxy", 10_000, 500.0, 0.15, 3.0, 0.5, 0.2, "barnes_hut"),
    "cluster_1m_direct": ("cluster_fast", 1_000_000, 300.0, 0.05, 1.0, 0.0, 0.02, "direct"),
    "boids_2m": ("boids", 2_000_000, 500.0, 0.0, 0.0, 0.0, 1.0 / 60.0, "boids"),
}


PROFILE_ROUND = "r04"


def csrc_sha():
    """Hash of the kernel sources: a committed PMC summary only describes the kernels it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "3d-spatial-sim-for-boid-and-nbody_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/, collected with scripts/gpu_pmc.sh): (2*FETCH_SIZE + WRITE_SIZE) KiB,
    i.e. with the gfx950 read-side correction MI355X_MICROARCH.md prescribes.  The counters cannot be
    read inside this process, so the figure comes from the profile - and only if that profile was taken
    from the kernel sources in this tree (stamp `_csrc_sha`); otherwise None."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{workload}_pmc_summary.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        if doc.get("_csrc_sha") != csrc_sha():
            return None
        d = doc[kernel_key]
        return {"bytes": d["hbm_bytes_fetch_x2"], "bytes_uncorrected": d["hbm_bytes_raw"],
                "source": os.path.relpath(path, ROOT), "csrc_sha": doc["_csrc_sha"]}
    except (OSError, KeyError, ValueError):
        return None


def issue_floor(cycles_per_visit, asked_share, every_wave, forced):
    s = 1.0 if (every_wave or forced == "f64") else (0.0 if forced == "f32" else float(asked_share))
    floor = 38.0 + s * (82.0 - 38.0)
    return {"cycles_per_wave_visit_per_simd": cycles_per_visit, "float64_visit_cycles": 82, "fp32_visit_cycles": 38,
            "float64_share_of_waves": round(s, 4), "floor_cycles": floor, "frac": floor / cycles_per_visit}


def make_ic(dist_name, n, R, G):
    from tools.presets import generate_distribution
    np.random.seed(42)
    if dist_name == "cluster_fast":
        # Plummer positions as the reference's `cluster` (tools/presets.py:350-365); its per-body
        # Python velocity loop takes minutes at 1 M, so velocities are drawn vectorised here
        # (same distribution, different random stream - the force kernel does not care).
        a = R * 0.3
        u = np.random.uniform(0, 1, n)
        r = np.clip(a / np.sqrt(u ** (-2 / 3) - 1), 0, R * 1.5)
        phi = np.random.uniform(0, 2 * np.pi, n)
        ct = np.random.uniform(-1, 1, n)
        st = np.sqrt(1 - ct ** 2)
        pos = np.stack([r * st * np.cos(phi), r * ct, r * st * np.sin(phi)], 1)
        sigma = np.sqrt(G * n * 0.001 / (6 * a)) * (1 + (r / a) ** 2) ** -0.25
        vel = np.random.normal(0, 1, (n, 3)) * sigma[:, None]
        return np.ascontiguousarray(pos), np.ascontiguousarray(vel), np.ones(n)
    return generate_distribution(dist_name, n, R, G)


_THREADS = {}


def cpu_baseline(p, v, m, theta, G, eps, dt, method, budget_s=25.0, strict_too=False):
    """Oracle timed on host cores for a bounded sample (about 10-30 s of CPU work)."""
    from oracle import pyref
    try:  # local -march=native build of the fast variant; fall back to the shipped one
        path = pyref.build(fast=True, native=True, out_dir="/tmp")
        L = pyref.lib(path=path)
    except Exception:
        L = pyref.lib(fast=True)
    n = len(p)

    def pick_threads(run_once):
        """The box may expose more hardware threads than it lets us use (a 1-GPU box of this pool shows
        256 and schedules ~16): time one pass per candidate count and keep the fastest.  Done once per
        process (the 10 M-body sample reuses the count found at 1 M: a pass there takes half a minute)."""
        if _THREADS.get("best"):
            L.nbref_set_num_threads(_THREADS["best"])
            return _THREADS["best"]
        most = int(L.nbref_num_threads())
        best, best_t = most, None
        for c in sorted({c for c in (16, 32, 64, most) if c <= most}):
            L.nbref_set_num_threads(c)
            t0 = time.perf_counter()
            run_once()
            t = time.perf_counter() - t0
            if best_t is None or t < best_t:
                best, best_t = c, t
        L.nbref_set_num_threads(best)
        _THREADS["best"] = best
        return best

    if method == "direct":
        ns = min(n, 65536)
        ps, ms_ = np.ascontiguousarray(p[:16384]), np.ascontiguousarray(m[:16384])
        cores = pick_threads(lambda: pyref.direct_forces(ps, ms_, G, eps, L=L))
        t0 = time.perf_counter()
        pyref.direct_forces(np.ascontiguousarray(p[:ns]), np.ascontiguousarray(m[:ns]), G, eps, L=L)
        t = time.perf_counter() - t0
        per_step_full = t * (n / ns) ** 2  # O(N^2) extrapolation from the sample
        return {"value": n / per_step_full, "unit": "body-steps/s", "cores": cores, "kind": "port",
                "sample": f"direct N^2 forces on the first {ns} bodies, {t:.2f}s, scaled by (N/{ns})^2"}
    st = pyref.BHStepper(p, v, m, theta, G, eps, 1.0, cap=pyref.UNCAPPED, rows=4 * n + 4096, L=L)
    st.step(dt)  # first step also pays first-touch of the node arrays: not timed
    cores = pick_threads(lambda: st.step(dt))
    st.phase_s[:] = 0
    t0 = time.perf_counter()
    st.step(dt)
    first = time.perf_counter() - t0
    steps = min(8, int(budget_s / max(first, 1e-3)))
    if steps < 1:  # one step already fills the budget (10 M bodies): it is the sample
        steps, t = 1, first
    else:
        st.phase_s[:] = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step(dt)
        t = time.perf_counter() - t0
    ph = st.phase_s / steps
    out = {"value": n * steps / t, "unit": "body-steps/s", "cores": cores, "kind": "port",
           "sample": f"{steps} full steps of the same {n}-body workload after 1 warm-up step "
                     f"(serial build {ph[2]:.2f}s + {cores}-thread walk {ph[3]:.2f}s per step; -O3 -ffast-math)"}
    if strict_too:
        # "and report a strict -O3 number" - the IEEE build (no fast-math, no contraction: the build
        # that reproduces the reference's goldens bit for bit), same threads, one timed step after one warm-up
        try:
            Ls = pyref.lib(path=pyref.build(fast=False, out_dir="/tmp"))
        except Exception:
            Ls = pyref.lib(fast=False)
        Ls.nbref_set_num_threads(cores)
        ss = pyref.BHStepper(p, v, m, theta, G, eps, 1.0, cap=pyref.UNCAPPED, rows=4 * n + 4096, L=Ls)
        ss.step(dt)
        ks = max(1, min(3, int(8.0 / max(first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(ks):
            ss.step(dt)
        ts = time.perf_counter() - t0
        out["strict_value"] = n * ks / ts
        out["strict_sample"] = f"{ks} steps, strict IEEE build (-O2 -ffp-contract=off, no fast-math), {cores} threads"
    return out


def frame_rates(sim, dt, n, substeps=5, frames=6):
    """The reference's per-frame call pattern (tools/record.py:821-832: `substeps` x step, compute_colors, get_positions,
    get_colors - the boundary hands host buffers over), timed end to end INCLUDING the device-to-host copies: raw
    float32 frames (24 B per body) and the device-side int16 delta codec (12 B per body).  Never `value`."""
    out = {"substeps": substeps, "frames_timed": frames}
    for name in ("raw", "delta_i16"):
        try:
            if name == "delta_i16":
                sim.compute_colors(15.0)
                sim.frame_keyframe()
            t0 = time.perf_counter()
            for _ in range(frames):
                sim.step_many(dt, substeps)
                sim.compute_colors(15.0)
                if name == "raw":
                    sim.get_positions()
                    sim.get_colors()
                else:
                    sim.frame_delta()
            t = (time.perf_counter() - t0) / frames
            out[name] = {"ms_per_frame": 1e3 * t, "body_steps_per_s": n * substeps / t}
        except Exception as ex:  # a measurement extra must not cost the bench line
            out[name] = {"error": repr(ex)}
    return out


def bench_boids(args, n, dt):
    """BASELINE config 5: boids/flock.py, 2 M boids, reference constants, dt = 1/60 (one GPU)."""
    import torch
    from boids import Flock
    from oracle import pyref
    with contextlib.redirect_stdout(sys.stderr):
        fl = Flock(n, seed=42)
    p0, v0, c0 = fl.positions.copy(), fl.velocities.copy(), fl.colors.copy()

    def fence():
        fl.sync()
        torch.cuda.synchronize()

    if args.presteps:  # (profiling the state flocks reach: the timed region starts after this many steps)
        fl.update(dt, args.presteps)
    fl.update(dt, args.warmup)
    fence()
    t0 = time.perf_counter()
    fl.update(dt, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    fl.enable_timers(True)
    fl.timers(reset=True)
    fl.update(dt, args.steps)
    fence()
    tm = fl.timers(reset=True)
    k = max(1, tm["steps"])
    sweep_ms = tm["sweep_ms"] / k
    info = fl.grid_info()
    # algorithmic bytes of the sweep kernel per boid: own AoS records 96 r + SoA state 76 w; per (y,z)
    # row of neighbour cells (9 of them) one 8-byte occupancy entry, and for a non-empty row two 4-byte
    # run bounds; candidates x 32 B position records (+64 B velocity/colour for the ~1 in range)
    occ_frac = info["occupied"] / info["num_cells"]
    cand = 27.0 * n / info["num_cells"]
    row_occ = 1.0 - (1.0 - occ_frac) ** 3
    alg = n * (96 + 76 + 9 * 8 + 9 * row_occ * 8 + cand * 32.0 + 1.0 * 64.0)
    model_gbs = alg / (sweep_ms * 1e-3) / 1e9
    # The model above counts every candidate record as a memory access; most of them are served by the caches
    # (counter traffic of the sweep is less than half of it).  `achieved` is therefore stated from the
    # COMPULSORY bytes - each boid's own records in (96) and its new state out (76) plus the occupancy table
    # once - which is what HBM has to move however the neighbours are found; the model rate rides along.
    compulsory = n * (96 + 76) + info["num_cells"] // 32 * 8 + info["occupied"] * 4
    ach = compulsory / (sweep_ms * 1e-3) / 1e9
    tr = pmc_traffic("boids_2m", "k_flock<true") if n == 2_000_000 else None
    out = {"metric": "boid-steps/sec (boids sep/align/cohesion sweep)", "value": n * args.steps / elapsed,
           "unit": "boid-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic (reference Flock ICs, seed 42)",
           "config": {"workload": "boids_2m", "boids": n, "bounds": 500.0, "perception_radius": 5.0, "dt": dt,
                      "grid_dim": info["grid_dim"], "num_cells": info["num_cells"]},
           "phase_ms": {key: tm[key] / k for key in ("sort_ms", "table_ms", "sweep_ms")},
           "roofline": {"bound": "hbm", "kernel": "k_flock", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS,
                        "traffic": tr["bytes"] if tr else None, "traffic_from_profile": tr,
                        "traffic_gbs": (tr["bytes"] / (sweep_ms * 1e-3) / 1e9) if tr else None,
                        "alg_bytes_per_launch": compulsory, "model_bytes_with_candidates": alg,
                        "model_gbs_with_candidates": model_gbs,
                        "kernel_ms": sweep_ms, "occupied_cells": info["occupied"]}}
    def candidates_per_boid():
        """mean number of candidates a boid tests: boids of the 27 cells around its own (itself included)"""
        cells = np.bincount(fl.cell_indices(), minlength=info["num_cells"]).astype(np.float64)
        d = info["grid_dim"]
        box = cells.reshape(d, d, d)  # cell index = x + y d + z d^2 (flock.py:41-43)
        for ax in range(3):
            pad = np.pad(box, [(1, 1) if a == ax else (0, 0) for a in range(3)])
            sl = [slice(None)] * 3
            acc = np.zeros_like(box)
            for o in range(3):
                sl[ax] = slice(o, o + d)
                acc += pad[tuple(sl)]
            box = acc
        per_cell = box.reshape(-1)  # candidates of a boid of that cell
        order = np.argsort(per_cell, kind="stable")
        cum = np.cumsum(cells[order]) / n  # share of the boids with at most that many candidates
        qs = {f"p{int(100 * f)}": float(per_cell[order][np.searchsorted(cum, f)]) for f in (0.5, 0.9, 0.99)}
        qs["max"] = float(per_cell[cells > 0].max())
        return float((cells * per_cell).sum() / n), qs

    out["config"]["candidates_per_boid"] = round(candidates_per_boid()[0], 2)
    if args.steady_steps and not args.presteps:
        # [r4] the state flocks REACH: the reference runs there, and the sweep costs twice what it
        # costs on the uniform initial state - cells empty out (occupied cells halve) and the rest fill up
        done = args.warmup + 2 * args.steps
        fl.enable_timers(False)
        fl.update(dt, max(0, args.steady_steps - done))
        fence()
        t0 = time.perf_counter()
        fl.update(dt, args.steps)
        fence()
        el2 = time.perf_counter() - t0
        fl.enable_timers(True)
        fl.timers(reset=True)
        fl.update(dt, args.steps)
        fence()
        tm2 = fl.timers(reset=True)
        k2 = max(1, tm2["steps"])
        info = fl.grid_info()
        comp2 = n * (96 + 76) + info["num_cells"] // 32 * 8 + info["occupied"] * 4
        sw2 = tm2["sweep_ms"] / k2
        out["steady_state"] = {"after_steps": max(done, args.steady_steps), "ms_per_step": 1e3 * el2 / args.steps,
                               "value": n * args.steps / el2, "unit": "boid-steps/s",
                               "phase_ms": {key: tm2[key] / k2 for key in ("sort_ms", "table_ms", "sweep_ms")},
                               "occupied_cells": info["occupied"], "candidates_per_boid": round(candidates_per_boid()[0], 2),
                               "candidates_per_boid_quantiles": candidates_per_boid()[1],
                               "roofline": {"bound": "hbm", "kernel": "k_flock", "achieved": comp2 / (sw2 * 1e-3) / 1e9,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": comp2 / (sw2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "alg_bytes_per_launch": comp2, "kernel_ms": sw2, "traffic": None}}
    if not args.no_cpu_baseline:
        try:
            L = pyref.lib(path=pyref.build(fast=True, native=True, out_dir="/tmp"))
        except Exception:
            L = pyref.lib(fast=True)
        st = pyref.FlockStepper(p0, v0, c0, pyref.boids_params(), use_numpy_argsort=True, L=L)
        st.step(dt)
        most, best, best_t = int(L.nbref_num_threads()), None, None
        for c in sorted({c for c in (16, 32, 64, most) if c <= most}):  # see cpu_baseline(): fastest thread count
            L.nbref_set_num_threads(c)
            t0 = time.perf_counter()
            st.step(dt)
            t = time.perf_counter() - t0
            if best_t is None or t < best_t:
                best, best_t = c, t
        L.nbref_set_num_threads(best)
        t0 = time.perf_counter()
        ksteps = 5
        for _ in range(ksteps):
            st.step(dt)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * ksteps / t, "unit": "boid-steps/s", "cores": int(L.nbref_num_threads()),
                               "kind": "port", "sample": f"{ksteps} full Flock.update steps of the same {n}-boid "
                               "workload after 1 warm-up (np.argsort + serial cell lists + OpenMP sweep)"}
    print(json.dumps(out), flush=True)


def bench_boids_slabs(args, n, dt, world, rank, dev):
    """BASELINE config 5 over N GPUs (strong scaling: the same 2 M boids, x-slabs of cell planes with a one-cell
    halo, one neighbour exchange per step; boids/sharded.py)."""
    import torch
    import torch.distributed as dist
    from boids.flock import generate_initial_state
    from boids.sharded import HipSlabEngine, SlabFlock
    from nbody.sharded import DistComm
    import config.boids as bcfg
    params = np.array([float(bcfg.BOIDS[k]) for k in bcfg.PARAM_ORDER], dtype=np.float64)
    np.random.seed(42)
    pos, vel, col = generate_initial_state(n, float(bcfg.BOIDS["bounds"]), float(bcfg.BOIDS["max_speed"]))
    with contextlib.redirect_stdout(sys.stderr):
        eng = HipSlabEngine(pos, vel, col, params, rank, world, device=dev)
    del pos, vel, col
    fl = SlabFlock(eng, rank, world, DistComm(dist, eng.device) if world > 1 else None)

    def fence():
        eng.wait()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    fl.step(dt, args.warmup)
    fence()
    t0 = time.perf_counter()
    fl.step(dt, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    return {"metric": "boid-steps/sec (boids sep/align/cohesion sweep)", "value": n * args.steps / elapsed,
            "unit": "boid-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic (reference Flock ICs, seed 42)",
            "config": {"workload": "boids_2m", "boids": n, "bounds": 500.0, "perception_radius": 5.0, "dt": dt,
                       "parallelism": f"x{world}: x-slabs of cell planes, one-cell halo + migrants in one all-to-all-v per step"},
            "exchange": {"rows_sent_last_step_rank0": int(eng.sent_rows), "row_bytes": 80}}


def check_owner_mode(dist, world, rank, G, eps, theta, dt, n=262_144, steps=3, tol=1e-7):
    """Owner mode ("let") against the replicated-tree mode ("rows", bit-identical to one GPU) on a small galaxy, over
    the process group of this very run.  Every rank returns the same verdict (the error is all-reduced).  Both walk the
    same global octree; in the default force precision the fp32 waves' sums associate differently (which 64 bodies form
    a wave differs) and each rank decides for its own waves: ~1e-9 after 3 steps, bound 1e-7."""
    import torch
    from nbody.sharded import create_sharded_simulation
    from tools.presets import generate_distribution
    res = {"ok": False, "bodies": n, "steps": steps, "tolerance": tol}
    err = float("inf")
    try:
        np.random.seed(7)
        p, v, m = generate_distribution("galaxy", n, 800.0, G)
        rows = create_sharded_simulation(p, v, m, G, eps, 1.0, theta, mode="rows")
        rows.step(dt, steps)
        rows.engine.sim.sync()
        ref = rows.engine.sim.get_positions_f64()
        rows.engine.sim.close()
        let = create_sharded_simulation(p, v, m, G, eps, 1.0, theta, mode="let")
        let.step(dt, steps)
        got, _ = let.gather_state()
        let.engine.sim.close()
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
    except Exception as ex:  # noqa: BLE001 - any failure of the untried mode means: use the tried one
        res["error"] = repr(ex)[:300]
    worst = torch.tensor([err if np.isfinite(err) else 1e30], dtype=torch.float64, device="cuda")
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    res["max_rel_position_diff"] = float(worst.item())
    res["ok"] = bool(res["max_rel_position_diff"] <= tol)
    return res


def measure_nbody(args, workload, world, rank, dev, use_dist, steps, warmup, cpu_budget_s=25.0):
    """One N-body workload: K timed steps (barrier + synchronize on both sides, max over ranks), then at
    N = 1 a second pass with per-phase HIP events, the counted walk, and the CPU port beside it."""
    import torch
    dist_name, per_gpu, R, G, eps, theta, dt, method = WORKLOADS[workload]
    if args.bodies_per_gpu and workload == args.workload:
        per_gpu = args.bodies_per_gpu
    if args.theta is not None:
        theta = args.theta
    if args.dt is not None and workload == args.workload:
        dt = args.dt
    # default workload: weak scaling (per_gpu bodies per rank).  Strong scaling (the same bodies in
    # total, sharded) for the direct N^2 kernel, whose work per body grows with N, and for
    # BASELINE config 4, which is "10 M bodies across the GPUs of one node".
    strong = method != "barnes_hut" or workload == "collision_10m_bh"
    n_total = per_gpu if strong else per_gpu * world
    p, v, m = make_ic(dist_name, n_total, R, G)

    from nbody import gpu_backend as gb
    # N > 1, Barnes-Hut: "let" = owned key ranges + exchanged locally essential trees (BASELINE north_star's form,
    # per-rank cost does not grow with N); "rows" = replicated tree, bit-identical to one GPU (the exact mode)
    shard_mode = os.environ.get("NBMI_SHARD_MODE", "let")
    mode_check = None
    with contextlib.redirect_stdout(sys.stderr):  # backend banners must not pollute the one JSON line
        if use_dist:
            import torch.distributed as dist
            from nbody.sharded import create_sharded_simulation
            if method == "barnes_hut" and shard_mode == "let" and world > 1 and "NBMI_SHARD_MODE" not in os.environ:
                # owner mode has never run over RCCL on >= 2 real GPUs.  Before anything is timed, the ranks
                # of THIS run step a small system in both modes and compare: owner mode must stay within its stated
                # tolerance of the bit-exact replicated-tree mode, or the timed run falls back to that mode.
                mode_check = check_owner_mode(dist, world, rank, G, eps, theta, dt)
                if not mode_check["ok"]:
                    shard_mode = "rows"
            sharded = create_sharded_simulation(p, v, m, G, eps, 1.0, theta, mode=shard_mode, method=method)
            sim = sharded.engine.sim
            if method == "barnes_hut" and args.force_precision:
                sim.set_force_precision(args.force_precision)
            step = lambda k: sharded.step(dt, k)  # noqa: E731
            del p, v, m  # every rank generated the whole system to pick its share; only the share stays
        else:
            sim = (gb.HIPBarnesHutSimulation(p, v, m, G, eps, 1.0, theta, device=dev) if method == "barnes_hut"
                   else gb.HIPDirectSimulation(p, v, m, G, eps, 1.0, device=dev))
            step = lambda k: sim.step_many(dt, k)  # noqa: E731
            if method == "barnes_hut" and args.force_precision:
                sim.set_force_precision(args.force_precision)

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    step(warmup)
    fence()
    t0 = time.perf_counter()
    step(steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    par = "single GPU"
    if world > 1 or use_dist:
        par = {"rows": f"x{world}: key-range shards, replicated state and tree, all-gather of 64-B rows",
               "let": f"x{world}: key-range owners, body migration (all-to-all-v), one global octree cut into the ranks' pieces, "
                      "all-to-all-v of the locally essential part of every piece",
               }.get(shard_mode, shard_mode) if method == "barnes_hut" else \
            f"x{world}: body-index shards of the all-pairs kernel, all-gather of 64-B rows"
    out = {
        "metric": f"body-steps/sec (N-body Barnes-Hut, theta={theta:g})" if method == "barnes_hut"
                  else "body-steps/sec (N-body direct N^2)",
        "value": n_total * steps / elapsed,
        "unit": "body-steps/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": ("f64 pair forces for the waves of 64 bodies whose densest quarter has G*rho*dt^2 > tau - and for every wave "
                  "while most of the system qualifies (entered above a third of the waves, left below a quarter; library default "
                  "'auto') -, f32 pair forces with f64 sums for the others; f64 opening-test ties, f64 state/keys") if method == "barnes_hut"
                 else "f32 pair forces, f64 sums/state",
        "data": "synthetic (reference IC generator restated, seed 42)",
        "config": {"workload": workload, "distribution": dist_name.replace("_fast", ""),
                   "bodies_per_gpu": n_total // world if strong else per_gpu, "bodies_total": n_total,
                   "theta": theta, "dt": dt, "G": G, "softening": eps, "spawn_radius": R, "method": method,
                   "force_precision": None if method != "barnes_hut" else (args.force_precision or "auto"),
                   "parallelism": par},
    }

    if mode_check is not None:
        out["config"]["owner_mode_check"] = mode_check
    if use_dist and method == "barnes_hut" and shard_mode == "let" and rank == 0:
        e = sharded.engine
        share, all64 = sim.force_precision_share()
        out["exchange"] = {"bytes_sent_per_step_rank0": int(e.wire_bytes), "bodies_migrated_last_step_rank0": int(e.migrated),
                           "received_tree_rows_rank0": [int(c) for c in e.let_counts], "owned_bodies_rank0": int(sim.n),
                           # rank 0's own waves that asked for float64 forces; whether the SYSTEM-WIDE vote made every wave float64
                           "float64_wave_share_rank0": round(float(share), 4), "all_waves_float64": bool(all64)}
    if world == 1 and rank == 0 and not use_dist:
        # second pass with per-phase HIP events on the library stream (same K steps)
        sim.enable_timers(True)
        sim.timers(reset=True)
        step(steps)
        fence()
        tm = sim.timers(reset=True)
        sim.enable_timers(False)
        k = max(1, tm["steps"])
        walk_ms = tm["walk_ms"] / k
        out["phase_ms"] = {key: tm[key] / k for key in ("keys_ms", "sort_ms", "tree_ms", "walk_ms")}
        if method == "barnes_hut":
            sim.accelerations()  # one counted walk on the final state
            wc = sim.walk_counters()
            ts = sim.tree_stats()
            groups = max(1, (n_total + 63) // 64)
            # algorithmic bytes of the walk = per body its state in and out (BODY_BYTES_WALK) plus
            # its node visits x the node record, amortised over the 64 bodies that share one fetch.
            alg_bytes = wc["lane_visits"] / 64.0 * NODE_BYTES + n_total * BODY_BYTES_WALK
            # what this lock-step design actually requests: one record per wave-level visit (the union of
            # its 64 bodies' visits, ~2x the above)
            req_bytes = wc["wave_visits"] * NODE_BYTES + n_total * BODY_BYTES_WALK
            ach = alg_bytes / (walk_ms * 1e-3) / 1e9
            tr = pmc_traffic(workload, "k_walk<true") if not args.bodies_per_gpu and args.theta is None else None
            cyc = walk_ms * 1e-3 * 2.4e9 * 1024 / max(1, wc["wave_visits"])
            out["roofline"] = {"bound": "hbm", "kernel": "k_walk", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": tr["bytes"] if tr else None, "traffic_from_profile": tr,
                               "alg_bytes_per_launch": alg_bytes, "kernel_ms": walk_ms,
                               "requested_bytes_per_launch": req_bytes,
                               "requested_gbs": req_bytes / (walk_ms * 1e-3) / 1e9,
                               "wave_visits_per_group": wc["wave_visits"] / groups,
                               "lane_visits_per_body": wc["lane_visits"] / n_total,
                               "interactions_per_s": wc["lane_accepts"] / (walk_ms * 1e-3),
                               "tie_visits_redecided_f64": wc["band_visits"],
                               # second reading: this kernel is not HBM-bound.  What it can be held to is
                               # the vector-issue time of its visits: a float64 visit is 17 float64-rate + 3 fp32-rate
                               # instructions + v_rsq_f32 = 82 cycles of its SIMD at the architectural rates (4 / 2 / 8),
                               # an fp32 visit 38; 1024 SIMDs at 2.4 GHz.  The float64 share of the VISITS is taken as the
                               # share of the waves (dense waves visit more: the floor is a little higher than this).
                               "issue_floor": issue_floor(cyc, *sim.force_precision_share(), args.force_precision),
                               "lane_efficiency": wc["lane_visits"] / max(1, 64 * wc["wave_visits"]),
                               "num_nodes": ts["num_nodes"], "max_depth": ts["max_depth"],
                               "jumps_per_group": wc["jumps"] / groups,
                               "xcd_visit_share": [round(x / max(1, wc["wave_visits"]), 4) for x in wc["xcd_visits"]]}
        else:
            flops = 20.0 * n_total * n_total
            ach = flops / (walk_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "fp32-valu", "kernel": "k_direct", "achieved": ach, "peak": 157.3,
                               "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None, "kernel_ms": walk_ms,
                               "interactions_per_s": n_total * n_total / (walk_ms * 1e-3)}
        if method == "barnes_hut":
            share, all64 = sim.force_precision_share()
            out["config"]["float64_wave_share"] = {"asked": round(share, 4), "every_wave_float64": all64}
        if method == "barnes_hut" and workload == args.workload:
            out["frame_pcie"] = frame_rates(sim, dt, n_total)
        if not args.no_cpu_baseline:
            sim.close()
            out["cpu_baseline"] = cpu_baseline(p, v, m, theta, G, eps, dt, method, budget_s=cpu_budget_s,
                                               strict_too=(workload == "galaxy_1m_bh"))
    with contextlib.suppress(Exception):
        sim.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="galaxy_1m_bh", choices=sorted(WORKLOADS))
    ap.add_argument("--bodies-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--presteps", type=int, default=0,
                    help="boids: untimed steps before the warm-up (profiles of the state flocks reach)")
    ap.add_argument("--steady-steps", type=int, default=1000,
                    help="boids: the line's steady_state object is measured after this many steps (0: skip)")
    ap.add_argument("--skip-10m", action="store_true",
                    help="default run only: leave out the second object (north_star's N = 10 M on this one GPU)")
    ap.add_argument("--dt", type=float, default=None,
                    help="override the workload's step (the `4k_galaxy_1m` preset steps galaxy_1m_bh at 0.05 / 5 = 0.01)")
    ap.add_argument("--force-precision", default=None, choices=["auto", "f32", "f64"],
                    help="pair-force arithmetic (default: the library's, auto; single-GPU Barnes-Hut workloads only)")
    ap.add_argument("--theta", type=float, default=None,
                    help="override the workload's opening angle (exploration; BASELINE's metric is theta = 0.5)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % max(1, ndev)  # == LOCAL_RANK on a real node; lets ranks share a GPU in rehearsals
    # NBMI_BENCH_FORCE_DIST=1: run the N > 1 code path (process group, shard engine, collective) with a
    # single rank - the RCCL smoke test a 1-GPU box allows
    use_dist = world > 1 or os.environ.get("NBMI_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run, or "
                 f"unset WORLD_SIZE and let bench.py start the ranks itself)")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(dev)
        backend = os.environ.get("NBMI_BENCH_BACKEND", "nccl")  # "gloo": 1-GPU rehearsal of the N>1 path
        import datetime
        patience = datetime.timedelta(seconds=int(os.environ.get("NBMI_BENCH_TIMEOUT_S", "300")))  # a stuck rank ends the run
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev), timeout=patience)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"

    method = WORKLOADS[args.workload][7]
    if method == "boids":
        if world > 1 or use_dist:
            out = bench_boids_slabs(args, WORKLOADS[args.workload][1], WORKLOADS[args.workload][6], world, rank, dev)
            if rank == 0:
                print(json.dumps(out), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            return
        return bench_boids(args, WORKLOADS[args.workload][1], WORKLOADS[args.workload][6])

    out = measure_nbody(args, args.workload, world, rank, dev, use_dist, args.steps, args.warmup)
    # The default line times BASELINE config 2 (1 M bodies: the configuration the metric is quoted on that
    # fits the "few minutes" budget with its CPU baseline).  north_star's target is quoted at N = 10 M, which
    # also fits one GPU: the same measurement for the config-4 input rides along as a second object.
    plain_default = (world == 1 and not use_dist and args.workload == "galaxy_1m_bh" and not args.bodies_per_gpu
                     and args.theta is None and args.dt is None and not args.skip_10m and not args.force_precision)
    if plain_default:
        out["north_star_10m"] = measure_nbody(args, "collision_10m_bh", 1, 0, dev, False, min(args.steps, 10),
                                              min(args.warmup, 2), cpu_budget_s=10.0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
