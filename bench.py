#!/usr/bin/env python3
"""Benchmark of the hot path: Barnes-Hut body-steps/sec at theta = 0.5 on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full timestep of the workload (bounds -> keys -> sort -> octree -> walk with
fused kick-drift) with the bodies already resident in HBM.  Default workload = BASELINE config 2
(galaxy, 1 M bodies per GPU, theta 0.5, dt 0.05, G 0.07, eps 1.5, R 800; synthetic IC from the
reference's generator restated in tools/presets.py, seed 42).  With N > 1 the work per GPU is
fixed (weak scaling): N x 1 M bodies, every rank builds the full octree and walks its own
key-range, one RCCL all-gather of the updated rows per step (nbody/sharded.py;
NBMI_SHARD_MODE=runs selects the experimental fixed-ownership run exchange).

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  roofline     - dominant kernel (k_walk): algorithmic bytes per launch / its mean duration,
                 measured with HIP events on the library's own stream (nbmi timers);
  cpu_baseline - the CPU oracle ("port" of the reference's CPU path, OpenMP, -O3 -ffast-math like
                 Numba fastmath) timed on this host for a bounded sample of the same workload.
"""
import argparse
import contextlib
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
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


def pmc_traffic(workload, kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
    same command (profiles/, collected with scripts/gpu_pmc.sh): (2*FETCH_SIZE + WRITE_SIZE) KiB,
    i.e. with the gfx950 read-side correction MI355X_MICROARCH.md prescribes.  None if absent."""
    path = os.path.join(ROOT, "profiles", f"r01_{workload}_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)[kernel_key]
        return {"bytes": d["hbm_bytes_fetch_x2"], "bytes_uncorrected": d["hbm_bytes_raw"],
                "source": os.path.relpath(path, ROOT)}
    except (OSError, KeyError, ValueError):
        return None


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


def cpu_baseline(p, v, m, theta, G, eps, dt, method, budget_s=25.0):
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
        256 and schedules ~16): time one pass per candidate count and keep the fastest."""
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
    t0 = time.perf_counter()
    st.step(dt)
    first = time.perf_counter() - t0
    steps = max(1, min(8, int(budget_s / max(first, 1e-3))))
    st.phase_s[:] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(dt)
    t = time.perf_counter() - t0
    ph = st.phase_s / steps
    return {"value": n * steps / t, "unit": "body-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full steps of the same {n}-body workload after 1 warm-up step "
                      f"(serial build {ph[2]:.2f}s + {cores}-thread walk {ph[3]:.2f}s per step; -O3 -ffast-math)"}


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
    ach = alg / (sweep_ms * 1e-3) / 1e9
    out = {"metric": "boid-steps/sec (boids sep/align/cohesion sweep)", "value": n * args.steps / elapsed,
           "unit": "boid-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic (reference Flock ICs, seed 42)",
           "config": {"workload": "boids_2m", "boids": n, "bounds": 500.0, "perception_radius": 5.0, "dt": dt,
                      "grid_dim": info["grid_dim"], "num_cells": info["num_cells"]},
           "phase_ms": {key: tm[key] / k for key in ("sort_ms", "table_ms", "sweep_ms")},
           "roofline": {"bound": "hbm", "kernel": "k_flock", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS,
                        "traffic": (pmc_traffic("boids_2m", "k_flock<true") or {}).get("bytes") if n == 2_000_000 else None,
                        "alg_bytes_per_launch": alg,
                        "kernel_ms": sweep_ms, "occupied_cells": info["occupied"]}}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="galaxy_1m_bh", choices=sorted(WORKLOADS))
    ap.add_argument("--bodies-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(dev)
        backend = os.environ.get("NBMI_BENCH_BACKEND", "nccl")  # "gloo": 1-GPU rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"

    dist_name, per_gpu, R, G, eps, theta, dt, method = WORKLOADS[args.workload]
    if args.bodies_per_gpu:
        per_gpu = args.bodies_per_gpu
    if args.theta is not None:
        theta = args.theta
    if method == "boids":
        assert world == 1, "boids run as replicas only (DESIGN.md section 6)"
        return bench_boids(args, per_gpu, dt)
    # default workload: weak scaling (per_gpu bodies per rank).  Strong scaling (the same bodies in
    # total, sharded) for the direct N^2 kernel, whose work per body grows with N, and for
    # BASELINE config 4, which is "10 M bodies across the GPUs of one node".
    strong = method != "barnes_hut" or args.workload == "collision_10m_bh"
    n_total = per_gpu if strong else per_gpu * world
    p, v, m = make_ic(dist_name, n_total, R, G)

    from nbody import gpu_backend as gb
    with contextlib.redirect_stdout(sys.stderr):  # backend banners must not pollute the one JSON line
        if use_dist:
            from nbody.sharded import create_sharded_simulation
            shard_mode = os.environ.get("NBMI_SHARD_MODE", "rows")  # "runs": experimental fixed-ownership exchange
            sharded = create_sharded_simulation(p, v, m, G, eps, 1.0, theta, mode=shard_mode, method=method)
            sim = sharded.engine.sim
            step = lambda k: sharded.step(dt, k)  # noqa: E731
        else:
            sim = (gb.HIPBarnesHutSimulation(p, v, m, G, eps, 1.0, theta, device=dev) if method == "barnes_hut"
                   else gb.HIPDirectSimulation(p, v, m, G, eps, 1.0, device=dev))
            step = lambda k: sim.step_many(dt, k)  # noqa: E731

    def fence():
        sim.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    step(args.warmup)
    fence()
    t0 = time.perf_counter()
    step(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    out = {
        "metric": f"body-steps/sec (N-body Barnes-Hut, theta={theta:g})" if method == "barnes_hut"
                  else "body-steps/sec (N-body direct N^2)",
        "value": n_total * args.steps / elapsed,
        "unit": "body-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32 forces, f64 state/keys",
        "data": "synthetic (reference IC generator restated, seed 42)",
        "config": {"workload": args.workload, "distribution": dist_name.replace("_fast", ""),
                   "bodies_per_gpu": n_total // world if strong else per_gpu, "bodies_total": n_total, "theta": theta, "dt": dt, "G": G,
                   "softening": eps, "spawn_radius": R, "method": method,
                   "parallelism": "single GPU" if world == 1 else
                                  (f"x{world}: fixed owners (initial key ranges), all-reduce max + all-gather of "
                                   "sorted 32-B runs per step, merged whole-system octree per rank"
                                   if os.environ.get("NBMI_SHARD_MODE", "rows") == "runs" else
                                   (f"x{world}: key-range shards, replicated state and tree, all-gather of 64-B rows"
                                    if method == "barnes_hut" else
                                    f"x{world}: body-index shards of the all-pairs kernel, all-gather of 64-B rows"))},
    }

    if world == 1 and rank == 0:
        # second pass with per-phase HIP events on the library stream (same K steps)
        sim.enable_timers(True)
        sim.timers(reset=True)
        step(args.steps)
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
            alg_bytes = wc["wave_visits"] * NODE_BYTES + n_total * BODY_BYTES_WALK
            ach = alg_bytes / (walk_ms * 1e-3) / 1e9
            tr = pmc_traffic(args.workload, "k_walk<true") if not args.bodies_per_gpu else None
            out["roofline"] = {"bound": "hbm", "kernel": "k_walk", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": tr["bytes"] if tr else None,
                               "traffic_detail": tr,
                               "alg_bytes_per_launch": alg_bytes, "kernel_ms": walk_ms,
                               "wave_visits_per_group": wc["wave_visits"] / max(1, (n_total + 63) // 64),
                               "lane_visits_per_body": wc["lane_visits"] / n_total,
                               "interactions_per_s": wc["lane_accepts"] / (walk_ms * 1e-3),
                               # second reading: this kernel is bound by vector-instruction issue and
                               # its per-wave load chain, not by HBM: 16 VALU instructions per wave-level visit at the
                               # measured 2.8 cycles per instruction per SIMD (scripts/ubench), 1024 SIMDs, 2.4 GHz
                               "valu_issue": {"valu_per_visit": 16, "cycles_per_valu": 2.8,
                                              "cycles_per_visit_per_simd": walk_ms * 1e-3 * 2.4e9 * 1024 / max(1, wc["wave_visits"]),
                                              "frac": 16 * 2.8 * wc["wave_visits"] / (walk_ms * 1e-3 * 2.4e9 * 1024)},
                               "lane_efficiency": wc["lane_visits"] / max(1, 64 * wc["wave_visits"]),
                               "num_nodes": ts["num_nodes"], "max_depth": ts["max_depth"],
                               "window_misses_per_group": {str(k): v / max(1, (n_total + 63) // 64)
                                                           for k, v in wc["window_misses"].items()},
                               "jumps_per_group": wc["jumps"] / max(1, (n_total + 63) // 64),
                               "xcd_visit_share": [round(v / max(1, wc["wave_visits"]), 4) for v in wc["xcd_visits"]]}
        else:
            flops = 20.0 * n_total * n_total
            ach = flops / (walk_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "fp32-valu", "kernel": "k_direct", "achieved": ach, "peak": 157.3,
                               "unit": "TFLOP/s", "frac": ach / 157.3, "traffic": None, "kernel_ms": walk_ms,
                               "interactions_per_s": n_total * n_total / (walk_ms * 1e-3)}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(p, v, m, theta, G, eps, dt, method)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
