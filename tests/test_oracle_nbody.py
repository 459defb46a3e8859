"""The CPU oracle (oracle/nbref.c) against golden vectors produced by the reference's own
functions (oracle/gen_golden.py).  Strict-IEEE build => bit-identical where the reference is
deterministic (tree arrays, accelerations, integration, colours)."""
import numpy as np
import pytest

from conftest import golden

TREES = ["tree_galaxy_256", "tree_galaxy_2048", "tree_collision_2048", "tree_cluster_2048"]


def _build(oracle, pos, mass, rows=None):
    b = oracle.compute_bounds(pos)
    nd = oracle.NodeArrays.for_bodies(len(pos), rows)
    nn = oracle.build_octree(pos, mass, b, nd)
    return b, nd, nn


@pytest.mark.parametrize("name", TREES)
def test_build_octree_bit_exact(oracle, name):
    g = golden(name)
    pos, mass = g["pos"], g["mass"]
    b, nd, nn = _build(oracle, pos, mass)
    assert b == float(g["bounds"])
    assert nn == int(g["num_nodes"])
    assert np.array_equal(nd.centers[:nn], g["node_centers"])
    assert np.array_equal(nd.half[:nn], g["node_half"])
    assert np.array_equal(nd.mass[:nn], g["node_mass"])
    assert np.array_equal(nd.com[:nn], g["node_com"])
    assert np.array_equal(nd.children[:nn], g["node_children"])
    assert np.array_equal(nd.body[:nn], g["node_body"])
    assert np.array_equal(nd.leaf[:nn].astype(bool), g["node_leaf"])


@pytest.mark.parametrize("name", TREES)
def test_forces_bit_exact(oracle, name):
    g = golden(name)
    pos, mass = g["pos"], g["mass"]
    b, nd, nn = _build(oracle, pos, mass)
    for theta, key in [(0.5, "acc_t050"), (0.95, "acc_t095")]:
        acc, st = oracle.compute_forces_barnes_hut(pos, mass, nd, nn, theta, float(g["G"]), float(g["eps"]), stats=True)
        assert np.array_equal(acc, g[key])
        assert st["dropped"] == 0


@pytest.mark.parametrize("name", TREES)
def test_cells_and_body_keys(oracle, name):
    """(level,key) cell set == reference's; the per-body 21-digit key has its leaf's path as prefix."""
    g = golden(name)
    pos, mass = g["pos"], g["mass"]
    b, nd, nn = _build(oracle, pos, mass)
    level, key = oracle.tree_cells(nd, nn)
    idx = np.lexsort((key, level))
    cells = np.stack([level[idx].astype(np.uint64), key[idx]], axis=1)
    assert np.array_equal(cells, g["cells"])
    assert int(level.max()) == int(g["max_depth"])
    assert np.array_equal(np.bincount(level, minlength=24), g["level_hist"])
    hi, lo = oracle.body_keys(pos, b)
    ll = g["leaf_level"].astype(np.uint64)
    assert np.array_equal(hi >> (np.uint64(63) - np.uint64(3) * ll), g["leaf_key"])


def test_edge_cases(oracle):
    g = golden("tree_edge_cases")
    for tag in ["n1", "n2", "lattice", "close_pairs", "heavy"]:
        pos, mass = g[tag + "_pos"], g[tag + "_mass"]
        b, nd, nn = _build(oracle, pos, mass, rows=8192)
        assert b == float(g[tag + "_bounds"])
        assert nn == int(g[tag + "_num_nodes"]), tag
        level, key = oracle.tree_cells(nd, nn)
        idx = np.lexsort((key, level))
        assert np.array_equal(np.stack([level[idx].astype(np.uint64), key[idx]], 1), g[tag + "_cells"]), tag
        acc = oracle.compute_forces_barnes_hut(pos, mass, nd, nn, 0.5, 1.0, 0.1)
        assert np.array_equal(acc, g[tag + "_acc"]), tag
        assert nd.mass[0] == float(g[tag + "_root_mass"])
        assert np.array_equal(nd.com[0], g[tag + "_root_com"])


def test_trajectory_2048_bit_exact(oracle):
    """100 steps of the record() CPU loop at N=2048: positions, velocities, colours, node counts."""
    g = golden("traj_galaxy_2048")
    st = oracle.BHStepper(g["pos_0"], g["vel_0"], g["mass"], float(g["theta"]), float(g["G"]), float(g["eps"]),
                          float(g["damping"]))
    nn = []
    for s in range(1, 101):
        nn.append(st.step(float(g["dt"])))
        if s in (1, 10, 100):
            assert np.array_equal(st.pos, g[f"pos_{s}"]), s
            assert np.array_equal(st.vel, g[f"vel_{s}"]), s
            assert np.array_equal(oracle.compute_colors_by_velocity(st.vel, 15.0), g[f"col_{s}"]), s
    assert np.array_equal(np.array(nn), g["num_nodes_per_step"])


def test_trajectory_10k_config1(oracle):
    """BASELINE config 1: quick_galaxy 10 K, theta 0.5, dt 0.2, 100 steps - the reference run."""
    from tools.presets import generate_distribution
    g = golden("traj_galaxy_10k")
    np.random.seed(42)
    p, v, m = generate_distribution("galaxy", 10_000, 500.0, 0.15)
    st = oracle.BHStepper(p, v, m, 0.5, 0.15, 3.0, 1.0)
    nn = [st.step(0.2) for _ in range(100)]
    assert np.array_equal(np.array(nn), g["num_nodes_per_step"])
    assert np.array_equal(st.pos, g["pos_100"])
    assert np.array_equal(st.vel, g["vel_100"])
    assert st.stats[2] == 0  # no dropped pushes (64-entry stack never overflowed)


def test_tree_10k_and_100k_facts(oracle):
    from tools.presets import generate_distribution
    for name, n in [("tree_galaxy_10k", 10_000), ("tree_galaxy_100k", 100_000)]:
        g = golden(name)
        np.random.seed(42)
        p, v, m = generate_distribution("galaxy", n, 500.0, 0.15)
        b, nd, nn = _build(oracle, p, m)
        assert b == float(g["bounds"])
        assert nn == int(g["num_nodes"])
        level, key = oracle.tree_cells(nd, nn)
        assert int(level.max()) == int(g["max_depth"])
        assert np.array_equal(np.bincount(level, minlength=24), g["level_hist"])
        if n == 10_000:
            acc = oracle.compute_forces_barnes_hut(p, m, nd, nn, 0.5, 0.15, 3.0)
            assert np.array_equal(acc, g["acc_t050"])
        else:
            acc = oracle.compute_forces_barnes_hut(p, m, nd, nn, 0.5, 0.15, 3.0)
            assert np.array_equal(acc[g["sample"]], g["acc_sample"])


def test_colors_ramp(oracle):
    g = golden("colors_ramp")
    assert np.array_equal(oracle.compute_colors_by_velocity(g["vel"], float(g["max_speed"])), g["colors"])


@pytest.mark.parametrize("name", ["direct_cluster_2048", "direct_galaxy_2048"])
def test_direct(oracle, name):
    g = golden(name)
    acc = oracle.direct_forces(g["pos"], g["mass"], float(g["G"]), float(g["eps"]))
    # golden = float64 NumPy evaluation of the same sum in a different order
    assert np.allclose(acc, g["acc"], rtol=1e-11, atol=1e-14)
    p, v = g["pos"].copy(), g["vel"].copy()
    oracle.direct_update(p, v, acc, float(g["dt"]), float(g["damping"]))
    assert np.allclose(p, g["pos_1"], rtol=1e-13) and np.allclose(v, g["vel_1"], rtol=1e-11, atol=1e-15)


def test_node_cap_quirk(oracle):
    """MAX_TREE_NODES emulation: with a tiny cap bodies are dropped but the call returns."""
    rng = np.random.RandomState(0)
    pos = rng.uniform(-10, 10, (500, 3))
    m = np.ones(500)
    b = oracle.compute_bounds(pos)
    nd = oracle.NodeArrays(4096)
    full = oracle.build_octree(pos, m, b, nd, cap=oracle.UNCAPPED)
    capped = oracle.build_octree(pos, m, b, nd, cap=200)
    attached = int(((nd.body[:200] >= 0) & (nd.leaf[:200] == 1)).sum())
    assert full > 500 and capped >= 200 and attached < 500


def test_visibility_points_matches_reference(oracle):
    """compute_visibility_points (simulation.py:403-434) on four camera poses: the mask the
    reference function produced is reproduced exactly."""
    g = golden("visibility_nbody")
    for k in range(4):
        cam = g[f"cam_{k}"]
        th, tv = g[f"tan_{k}"]
        mask = oracle.compute_visibility_points(g["pos"], cam[0:3], cam[3:6], cam[6:9], cam[9:12], float(th), float(tv),
                                                float(g["far"]))
        assert np.array_equal(mask, g[f"mask_{k}"])
        assert 0 < mask.sum() <= len(mask)


def test_reference_node_cap_at_config_4_size(oracle):
    """BASELINE config 4's input at its full size through the reference's own 8 M-node cap (simulation.py:35, 141, 176):
    what the reference's build_octree did with exactly this input was recorded when the survey ran the reference itself
    (collision, 10 M bodies, seed 42, bounds 2948.0) - it returns num_nodes = 12 602 054 (the
    counter keeps running past the cap), attaches 5 085 518 bodies and loses everything from index 5 000 000 on bar
    85 518.  The oracle's serial-insertion build with the cap reproduces those three facts, which pins it (initial
    conditions, bounds, octant rule, insertion order, cap semantics) against the reference AT 10 M bodies; the product
    does not emulate the cap (it builds the complete octree: 14.8 M nodes, tests/test_gpu_nbody.py), and parity at this
    size is defined against the uncapped restatement for that reason."""
    from tools.presets import generate_distribution
    np.random.seed(42)
    pos, _, m = generate_distribution("collision", 10_000_000, 2000.0, 0.08)
    b = oracle.compute_bounds(pos)
    assert abs(b - 2948.0) < 0.05
    nd = oracle.NodeArrays(oracle.MAX_TREE_NODES)  # min(8 M, 4 N) rows, as tools/record.py:795 allocates them
    nn = oracle.build_octree(pos, m, b, nd, cap=oracle.MAX_TREE_NODES)
    held = nd.body[(nd.body >= 0) & (nd.leaf == 1)]
    assert nn == 12_602_054
    assert len(held) == 5_085_518 and len(np.unique(held)) == len(held)
    lost = np.ones(len(pos), dtype=bool)
    lost[held] = False
    assert int(np.argmax(lost)) == 5_000_000 and int(lost.sum()) == 4_914_482
    assert int((~lost[5_000_000:]).sum()) == 85_518
