"""Owner mode over many steps: 8 virtual ranks (threads on one GPU, LetBarnesHut.step itself) against the single handle,
both with float64 forces, every 100 steps - bodies migrate, rank boundaries wander through the tree, the pieces are
re-cut every step.  python scripts/gpu_owner_soak.py [n] [world] [steps] [dist] [f64 | auto]
(auto [r4]: the system-wide "every wave float64" verdict of the ranks against the single handle's, through a transition)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-spatial-sim-for-boid-and-nbody_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    dist = sys.argv[4] if len(sys.argv) > 4 else "collision"
    mode = sys.argv[5] if len(sys.argv) > 5 else "f64"
    from nbody.gpu_backend import HIPBarnesHutSimulation
    from nbody.sharded import HipLetEngine, LetBarnesHut
    from tools.presets import generate_distribution
    from test_gpu_sharded_record import _ThreadComm, _run_ranks
    G, eps, theta, dt = 0.08, 2.0, 0.5, 0.1
    np.random.seed(5)
    pos, vel, mass = generate_distribution(dist, n, 600.0, G)
    single = HIPBarnesHutSimulation(pos, vel, mass, G, eps, 1.0, theta)
    single.set_force_precision(mode)
    comm = _ThreadComm(world)
    engines = [HipLetEngine(pos, vel, mass, G, eps, 1.0, theta, 0, r, world) for r in range(world)]
    for e in engines:
        e.sim.set_force_precision(mode)
    steppers = [LetBarnesHut(e, r, world, comm.bind(r)) for r, e in enumerate(engines)]
    done = 0
    while done < steps:
        k = min(100, steps - done)
        t0 = time.time()
        out = _run_ranks(steppers, comm, dt, k)
        single.step_many(dt, k)
        ref = single.get_positions_f64()
        done += k
        d = np.abs(out[0][0] - ref).max(axis=1) / np.abs(ref).max()
        sh = single.force_precision_share()
        print(json.dumps({"dist": dist, "n": n, "world": world, "mode": mode, "steps": done, "max_vs_single": float(d.max()),
                          "single_share_all64": [round(sh[0], 3), bool(sh[1])],
                          "ranks_all64": [int(e.sim.force_precision_share()[1]) for e in engines],
                          "ranks_share": [round(e.sim.force_precision_share()[0], 3) for e in engines],
                          "owned": [int(e.sim.n) for e in engines], "let_rows": [int(e.let_counts.sum()) for e in engines],
                          "migrated_last_step": [int(e.migrated) for e in engines], "wall_s": round(time.time() - t0, 2)}), flush=True)
    for e in engines:
        e.sim.close()


if __name__ == "__main__":
    main()
