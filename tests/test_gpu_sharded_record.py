"""GPU tests of the multi-GPU shard engine (as virtual shards on one GPU) and of record()."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_virtual_shards_bit_identical_to_single_handle(gpu):
    """Three handles on one GPU play three ranks: each integrates its key-range, rows are exchanged
    through device buffers exactly as nbody/sharded.py does (the all-gather itself replaced by
    torch.cat).  The result must equal the unsharded handle bit for bit."""
    import torch
    from nbody.gpu_backend import HIPBarnesHutSimulation
    from nbody.sharded import HipShardEngine, ShardedBarnesHut, shard_bounds
    g = golden("tree_collision_2048")
    n = 2001  # ragged
    pos, vel, mass = g["pos"][:n], g["vel"][:n], g["mass"][:n]
    G, eps = float(g["G"]), float(g["eps"])
    world = 3
    single = HIPBarnesHutSimulation(pos, vel, mass, G, eps, 1.0, 0.5)
    engines = [HipShardEngine(pos, vel, mass, G, eps, 1.0, 0.5, 0) for _ in range(world)]
    shards = [ShardedBarnesHut(e, n, r, world, dist=None) for r, e in enumerate(engines)]
    per = shards[0].per
    assert [(s.begin, s.end) for s in shards] == [shard_bounds(n, world, r)[1:] for r in range(world)]
    for _ in range(5):
        single.step(0.05)
        for s in shards:
            s.engine.step(0.05)
            s.engine.export_rows(s.mine)
        full = torch.cat([s.mine for s in shards], dim=0)
        assert full.shape == (per * world, 8)
        for s in shards:
            s.engine.import_rows(full, n)
    ref_p, ref_v = single.get_positions_f64(), single.get_velocities()
    for e in engines:
        assert np.array_equal(e.sim.get_positions_f64(), ref_p)
        assert np.array_equal(e.sim.get_velocities(), ref_v)
    # a sharded handle refuses multi-substep calls (the other ranks' rows would be missing)
    with pytest.raises(RuntimeError, match="sharded handle"):
        engines[0].sim.step_many(0.05, 2)
    # world_size 1 goes through the same class without any exchange
    one = ShardedBarnesHut(HipShardEngine(pos, vel, mass, G, eps, 1.0, 0.5, 0), n, 0, 1)
    one.step(0.05, 5)
    assert np.array_equal(one.engine.sim.get_positions_f64(), ref_p)


def test_record_writes_reference_format_and_resumes(gpu, tmp_path, oracle):
    from tools import record as rec
    from tools.presets import generate_distribution, get_preset_config
    cfg = get_preset_config("quick_galaxy")
    cfg.update(num_bodies=3000, theta=0.5, total_frames=60, substeps=2, session_name="t_rec")
    d = rec.record(cfg, root=tmp_path, quiet=True, seed=42)
    assert rec.get_completed_frames(d) == 60
    meta = rec.load_metadata(d)
    assert meta["num_bodies"] == 3000 and meta["substeps"] == 2 and "start_datetime" in meta
    p, c = rec.load_frame(d, 59)
    assert p.dtype == np.float32 and p.shape == (3000, 3) and c.dtype == np.float32 and c.shape == (3000, 3)
    with np.load(d / "frame_0000.npz") as f:
        assert sorted(f.files) == ["colors", "positions"]
    # state checkpoint every 50 frames, keys of the reference + masses
    assert (d / "state_0049.npz").exists() and not (d / "state_0099.npz").exists()
    with np.load(d / "state_0049.npz") as st:
        assert {"positions", "velocities"} <= set(st.files)
        assert st["positions"].shape == (3000, 3) and st["velocities"].dtype == np.float64
    # the frames follow the reference's CPU loop (dt = dt_per_frame / substeps)
    np.random.seed(42)
    ip, iv, im = generate_distribution("galaxy", 3000, 500.0, 0.15)
    ref = oracle.BHStepper(ip, iv, im, 0.5, 0.15, 3.0, 1.0)
    for _ in range(60 * 2):
        ref.step(0.1)
    err = np.abs(p - ref.pos).max() / np.abs(ref.pos).max()
    print("record frame 59 rel err vs oracle", err)
    assert err < 1e-4
    assert np.abs(c - oracle.compute_colors_by_velocity(ref.vel, 15.0)).max() < 5e-3
    # resume: drop the tail, continue from state_0049 -> same frames again
    for k in range(50, 60):
        (d / f"frame_{k:04d}.npz").unlink()
    cfg2 = dict(cfg)
    d2 = rec.record(cfg2, resume=True, root=tmp_path, quiet=True)
    assert d2 == d and rec.get_completed_frames(d) == 60
    p2, _ = rec.load_frame(d, 59)
    assert np.array_equal(p2, p)
    # raw -> .zstd conversion keeps the chain loadable
    try:
        rec._load_zstd()
    except RuntimeError:
        return
    assert rec.compress_recording(d) == 60
    q, _ = rec.load_frame(d, 59)
    assert np.abs(q - p).max() < 0.06  # int16 millis quantisation accumulates over the chain
