"""Device-side initial conditions: statistical parity of nbmi_create_generated with
the NumPy generator of tools/presets.py (itself bit-identical to the reference's, tests/golden/ic_pins)."""
import time

import numpy as np
import pytest
from scipy import stats

pytestmark = pytest.mark.gpu

N = 200_000


def _host(dist, R, G, seed=42):
    from tools.presets import generate_distribution
    np.random.seed(seed)
    return generate_distribution(dist, N, R, G)


def _device(dist, R, G, eps, seed=42, n=N):
    from tools.presets import generate_distribution_device
    sim = generate_distribution_device(dist, n, R, G, eps, seed=seed)
    return sim, sim.get_positions_f64(), sim.get_velocities(), sim.get_masses()


def _ks(a, b):
    return stats.ks_2samp(a, b).statistic


def _binned_mean(r, v, edges):
    idx = np.digitize(r, edges)
    return np.array([v[idx == k].mean() for k in range(1, len(edges))])


def test_galaxy_statistics_match_host_generator(gpu):
    R, G = 500.0, 0.15
    hp, hv, hm = _host("galaxy", R, G)
    sim, dp, dv, dm = _device("galaxy", R, G, 3.0)
    assert np.all(dm == 1.0) and np.isfinite(dp).all() and np.isfinite(dv).all()
    hr, dr = np.hypot(hp[:, 0], hp[:, 2]), np.hypot(dp[:, 0], dp[:, 2])
    # two-sample KS distance between 200 k-point samples of the same law is ~ 0.003; 0.01 rejects any formula slip
    assert _ks(hr, dr) < 0.01
    assert _ks(hp[:, 1], dp[:, 1]) < 0.01                       # disk thickness
    assert _ks(np.arctan2(hp[:, 2], hp[:, 0]), np.arctan2(dp[:, 2], dp[:, 0])) < 0.01
    # rotation curve: mean tangential speed per radial bin (counter-clockwise in XZ)
    def vtan(p, v):
        r = np.hypot(p[:, 0], p[:, 2])
        return (p[:, 0] * v[:, 2] - p[:, 2] * v[:, 0]) / r
    edges = np.quantile(hr, np.linspace(0.02, 0.98, 13))
    hc, dc = _binned_mean(hr, vtan(hp, hv), edges), _binned_mean(dr, vtan(dp, dv), edges)
    assert np.all(np.abs(dc - hc) < 0.01 * np.abs(hc).max() + 0.02 * np.abs(hc)) and hc.min() > 0
    # dispersions
    assert abs(dv[:, 1].std() / hv[:, 1].std() - 1) < 0.02
    assert _ks(hv[:, 1], dv[:, 1]) < 0.01
    assert np.abs(dv.mean(axis=0)).max() < 1e-10               # centre-of-mass velocity removed
    # determinism and seed dependence
    _, dp2, dv2, _ = _device("galaxy", R, G, 3.0)
    assert np.array_equal(dp, dp2) and np.array_equal(dv, dv2)
    _, dp3, _, _ = _device("galaxy", R, G, 3.0, seed=43)
    assert not np.array_equal(dp, dp3) and _ks(np.hypot(dp3[:, 0], dp3[:, 2]), dr) < 0.01
    # and it is a working simulation handle
    sim.step_many(0.05, 3)
    assert np.isfinite(sim.get_positions_f64()).all()
    assert sim.tree_stats()["num_nodes"] > N


def test_collision_statistics_match_host_generator(gpu):
    R, G = 2000.0, 0.08
    hp, hv, _ = _host("collision", R, G)
    _, dp, dv, _ = _device("collision", R, G, 6.0)
    half = N // 2
    for sl in (slice(0, half), slice(half, N)):
        assert abs(dp[sl, 0].mean() - hp[sl, 0].mean()) < 0.01 * R        # disk centres -+ separation/2
        assert abs(dp[sl, 1].mean() - hp[sl, 1].mean()) < 0.002 * R       # y offset of the second disk
        # approach speed +-collision_speed: a mean over 1e5 draws, compare within 5 standard errors
        se = np.sqrt((hv[sl, 0].var() + dv[sl, 0].var()) / half)
        assert abs(dv[sl, 0].mean() - hv[sl, 0].mean()) < 5 * se
        c_h, c_d = hp[sl] - hp[sl].mean(axis=0), dp[sl] - dp[sl].mean(axis=0)
        assert _ks(np.hypot(c_h[:, 0], c_h[:, 2]), np.hypot(c_d[:, 0], c_d[:, 2])) < 0.015
        # spin: first disk counter-clockwise, second clockwise (angular momentum about its own centre)
        lz_h = (c_h[:, 0] * (hv[sl, 2]) - c_h[:, 2] * (hv[sl, 0] - hv[sl, 0].mean())).mean()
        lz_d = (c_d[:, 0] * (dv[sl, 2]) - c_d[:, 2] * (dv[sl, 0] - dv[sl, 0].mean())).mean()
        assert np.sign(lz_h) == np.sign(lz_d) and abs(lz_d / lz_h - 1) < 0.03


def test_cluster_statistics_match_host_generator(gpu):
    R, G = 300.0, 0.05
    hp, hv, _ = _host("cluster", R, G)
    _, dp, dv, _ = _device("cluster", R, G, 1.0)
    assert _ks(np.linalg.norm(hp, axis=1), np.linalg.norm(dp, axis=1)) < 0.01
    assert _ks(np.linalg.norm(hv, axis=1), np.linalg.norm(dv, axis=1)) < 0.01
    for k in range(3):  # isotropy
        assert _ks(hp[:, k], dp[:, k]) < 0.01 and _ks(hv[:, k], dv[:, k]) < 0.01
    assert np.linalg.norm(dp, axis=1).max() <= R * 1.5 * (1 + 1e-12)
    assert np.abs(dv.mean(axis=0)).max() < 1e-10


def test_ten_million_bodies_generated_on_device(gpu):
    """BASELINE config 4's input without the host: time it, sanity-check it, step it."""
    from tools.presets import generate_distribution_device
    t0 = time.perf_counter()
    sim = generate_distribution_device("collision", 10_000_000, 2000.0, 0.08, 6.0, seed=42)
    sim.sync()
    t_gen = time.perf_counter() - t0
    sim.step_many(0.25, 2)
    st = sim.tree_stats()
    p = sim.get_positions()
    assert np.isfinite(p).all() and st["num_nodes"] > 10_000_000
    # same statistics as the NumPy generator at this size: 14.8 M nodes, depth 22
    assert abs(st["num_nodes"] / 14_816_834 - 1) < 0.01
    print(f"10 M collision bodies generated on the device in {1e3 * t_gen:.0f} ms "
          f"(handle creation included); nodes {st['num_nodes']}, depth {st['max_depth']}")


def test_record_with_device_ic(gpu, tmp_path):
    from tools import record as rec
    from tools.presets import get_preset_config
    cfg = get_preset_config("quick_galaxy")
    cfg.update(num_bodies=5000, theta=0.5, total_frames=4, substeps=1, session_name="t_dev_ic", device_ic=True)
    d = rec.record(cfg, root=tmp_path, quiet=True, seed=7)
    assert rec.get_completed_frames(d) == 4
    p, c = rec.load_frame(d, 3)
    assert p.shape == (5000, 3) and np.isfinite(p).all()


def test_generated_edge_cases(gpu):
    from nbody.gpu_backend import HIPBarnesHutSimulation, HIPDirectSimulation
    with pytest.raises(ValueError):
        HIPBarnesHutSimulation.generated("ring", 10, 100.0, 0.1, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="bad arguments"):
        HIPBarnesHutSimulation.generated("galaxy", -1, 100.0, 0.1, 1.0, 1.0)
    e = HIPBarnesHutSimulation.generated("galaxy", 0, 100.0, 0.1, 1.0, 1.0)
    e.step(0.1)
    assert e.get_positions().shape == (0, 3)
    one = HIPBarnesHutSimulation.generated("cluster", 1, 100.0, 0.1, 1.0, 1.0)
    one.step(0.1)
    assert np.isfinite(one.get_positions_f64()).all() and np.all(one.get_velocities() == 0.0)  # COM removed
    odd = HIPBarnesHutSimulation.generated("collision", 1001, 2000.0, 0.08, 6.0, 1.0)  # halves of 500 / 501
    p = odd.get_positions_f64()
    assert (p[:500, 0] < 0).mean() > 0.99 and (p[500:, 0] > 0).mean() > 0.99
    d = HIPDirectSimulation.generated("cluster", 3000, 300.0, 0.05, 1.0, 1.0, seed=5)
    d.step(0.02)
    assert np.isfinite(d.get_positions_f64()).all()


def test_fifty_million_body_preset_fits(gpu):
    """The reference's largest presets ("50 Million Star Galaxy": theta 1.5, G 0.04, eps 10, R 3000,
    dt 0.35; tools/presets.py:2479-2493) on one GPU: generated on the device, stepped, octree inside its
    row budget (the reference itself loses bodies beyond its 8 M-node cap)."""
    from tools.presets import generate_distribution_device
    n = 50_000_000
    sim = generate_distribution_device("galaxy", n, 3000.0, 0.04, 10.0, theta=1.5, seed=1)
    sim.sync()
    t0 = time.perf_counter()
    sim.step_many(0.35, 2)
    sim.sync()
    dt = (time.perf_counter() - t0) / 2
    st = sim.tree_stats()
    assert n < st["num_nodes"] < 1.7 * n
    p = sim.get_positions()
    assert p.shape == (n, 3) and np.isfinite(p).all()
    with pytest.raises(RuntimeError, match="bad arguments"):
        generate_distribution_device("galaxy", 100_000_001, 3000.0, 0.04, 10.0)
    print(f"50 M bodies, theta 1.5: {1e3 * dt:.1f} ms/step, {st['num_nodes']} nodes, depth {st['max_depth']}")
