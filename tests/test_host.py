"""Host-side logic that needs no GPU: IC generators, frame/state format, C-ABI exports,
package surface, loud failure without a device."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden, have_gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("dist,n,R,G", [("galaxy", 10_000, 500.0, 0.15), ("collision", 10_000, 2000.0, 0.08),
                                        ("cluster", 4096, 300.0, 0.05), ("galaxy", 2048, 500.0, 0.15),
                                        ("collision", 2048, 2000.0, 0.08), ("cluster", 2048, 300.0, 0.05),
                                        ("galaxy", 256, 500.0, 0.15), ("galaxy", 100_000, 500.0, 0.15)])
def test_ic_generators_match_reference(dist, n, R, G):
    from tools.presets import generate_distribution
    g = golden("ic_pins")
    np.random.seed(42)
    p, v, m = generate_distribution(dist, n, R, G)
    t = f"{dist}_{n}"
    assert np.array_equal(p[:64], g[t + "_pos_head"]) and np.array_equal(p[-64:], g[t + "_pos_tail"])
    assert np.array_equal(v[:64], g[t + "_vel_head"]) and np.array_equal(v[-64:], g[t + "_vel_tail"])
    assert _sha(p) == str(g[t + "_pos_sha"]) and _sha(v) == str(g[t + "_vel_sha"]) and _sha(m) == str(g[t + "_mass_sha"])


def test_unknown_distribution_raises():
    from tools.presets import generate_distribution
    with pytest.raises(ValueError):
        generate_distribution("torus", 10, 1.0, 1.0)


def test_presets():
    from tools.presets import PRESETS, get_preset_config
    q = get_preset_config("quick_galaxy")
    assert q["session_name"] == "quick_galaxy" and q["num_bodies"] == 100_000 and q["theta"] == 0.95
    assert q["dt_per_frame"] == 0.2 and q["substeps"] == 1 and q["softening"] == 3.0 and q["G"] == 0.15
    assert PRESETS["4k_galaxy_1m"]["spawn_radius"] == 800.0 and PRESETS["4k_galaxy_1m"]["G"] == 0.07
    assert PRESETS["extreme_10m_collision"]["softening"] == 6.0
    assert get_preset_config("nope") is None
    assert "session_name" not in PRESETS["quick_galaxy"]


def test_boids_initial_state_matches_reference():
    from boids.flock import generate_initial_state
    for tag in ("sparse", "dense", "walls"):
        g = golden("boids_" + tag)
        np.random.seed(42)
        p, v, c = generate_initial_state(len(g["pos_0"]), np.float64(g["bounds"]), np.float64(25.0))
        assert np.array_equal(p, g["pos_0"]) and np.array_equal(v, g["vel_0"]) and np.array_equal(c, g["col_0"])


def test_live_ic_generators_shapes():
    from nbody import simulation as sim
    np.random.seed(1)
    for name in ("galaxy", "spiral", "sphere", "collision"):
        p, v, m = sim._LIVE_ICS[name](1000, 500.0, 0.1)
        assert p.shape == (1000, 3) and v.shape == (1000, 3) and m.shape == (1000,)
        assert p.dtype == v.dtype == m.dtype == np.float64 and np.isfinite(p).all() and np.isfinite(v).all()
    p, v, m = sim._ic_uniform(100, 500.0, 0.1)
    assert np.abs(p).max() <= 400.0
    pg, vg, mg = sim._ic_galaxy(5000, 500.0, 0.1)
    assert (mg == 100.0).sum() == 5 and np.abs(pg[:, 1]).mean() < np.abs(pg[:, 0]).mean()  # thin in y


# ---- frame / state format ------------------------------------------------------------------
def test_load_reference_written_frames():
    from tools.record import load_frame
    g = golden("frame_codec")
    d = os.path.join(GOLDEN, "frames")
    p0, c0 = load_frame(d, 0)
    p1, c1 = load_frame(d, 1)
    assert p0.dtype == np.float32 and c0.dtype == np.float32 and p0.shape == (16, 3)
    assert np.array_equal(p0, g["p0"]) and np.array_equal(c0, g["c0"])
    assert np.array_equal(p1, g["p1"]) and np.array_equal(c1, g["c1"])


def test_save_frame_bytes_identical_to_reference(tmp_path):
    from tools.record import save_frame
    g = golden("frame_codec")
    save_frame(tmp_path, 0, g["pos64_0"], g["c0"])
    ref = open(os.path.join(GOLDEN, "frames", "frame_0000.npz"), "rb").read()
    mine = open(tmp_path / "frame_0000.npz", "rb").read()
    with np.load(tmp_path / "frame_0000.npz") as d:
        assert sorted(d.files) == ["colors", "positions"]
        assert d["positions"].dtype == np.float32 and d["colors"].dtype == np.float32
    # zip members carry no timestamps in np.savez -> byte identical
    assert mine == ref


def test_delta_quantisation_matches_reference_incl_wrap():
    from tools.record import delta_quantize
    g = golden("frame_codec")
    dp = delta_quantize(g["p1"], g["p0"])
    assert dp.dtype == np.int16 and np.array_equal(dp, g["dpos_i16"])
    assert np.array_equal(delta_quantize(g["c1"], g["c0"]), g["dcol_i16"])
    assert (np.abs((g["p1"] - g["p0"]) * 1000) > 32767).any()  # the wrap-around case is present


def test_zstd_container_roundtrip(tmp_path):
    from tools import record as rec
    try:
        rec._load_zstd()
    except RuntimeError:
        pytest.skip("libzstd not present")
    g = golden("frame_codec")
    blob1 = rec.compress_frame(g["p0"], g["c0"])
    assert blob1[0] == 1
    q0, k0 = rec.decompress_frame(blob1)
    assert np.array_equal(q0, g["p0"]) and np.array_equal(k0, g["c0"])
    blob2 = rec.compress_frame(g["p1"], g["c1"], g["p0"], g["c0"])
    assert blob2[0] == 2
    q1, k1 = rec.decompress_frame(blob2, g["p0"], g["c0"])
    assert np.array_equal(q1, g["dec_p1"]) and np.array_equal(k1, g["dec_c1"])
    with pytest.raises(ValueError):
        rec.decompress_frame(blob2)
    with pytest.raises(ValueError):
        rec.decompress_frame(b"")
    with pytest.raises(ValueError):
        rec.decompress_frame(bytes([7]) + blob1[1:])
    # a recording: raw frames -> compress_recording -> load_frame chain (delta back-chaining)
    rng = np.random.RandomState(0)
    frames = []
    p = rng.normal(0, 50, (32, 3)).astype(np.float32)
    c = rng.uniform(0, 1, (32, 3)).astype(np.float32)
    for i in range(7):
        rec.save_frame(tmp_path, i, p, c)
        frames.append((p.copy(), c.copy()))
        p = p + rng.normal(0, 0.1, p.shape).astype(np.float32)
    assert rec.get_completed_frames(tmp_path) == 7
    assert rec.compress_recording(tmp_path) == 7
    assert rec.get_completed_frames(tmp_path) == 7 and not (tmp_path / "frame_0003.npz").exists()
    q, k = rec.load_frame(tmp_path, 6)
    assert np.abs(q - frames[6][0]).max() < 2e-3 and np.abs(k - frames[6][1]).max() < 2e-3
    q0, _ = rec.load_frame(tmp_path, 0)
    assert np.array_equal(q0, frames[0][0])
    with pytest.raises(FileNotFoundError):
        rec.load_frame(tmp_path, 99)


def test_metadata_and_state_helpers(tmp_path):
    from tools import record as rec
    from tools.presets import get_preset_config
    cfg = get_preset_config("quick_galaxy")
    rec.save_metadata(tmp_path, cfg, 1000.0)
    m = rec.load_metadata(tmp_path)
    for k in ("num_bodies", "theta", "total_frames", "distribution", "dt_per_frame", "substeps", "G", "softening",
              "damping", "spawn_radius", "session_name", "start_time", "start_datetime"):
        assert k in m
    assert rec.find_latest_state(tmp_path, 100) == (None, -1)
    np.savez(tmp_path / "state_0049.npz", positions=np.zeros((2, 3)), velocities=np.zeros((2, 3)))
    np.savez(tmp_path / "state_0099.npz", positions=np.zeros((2, 3)), velocities=np.zeros((2, 3)))
    f, k = rec.find_latest_state(tmp_path, 80)
    assert k == 49 and f.name == "state_0049.npz"


# ---- C ABI ------------------------------------------------------------------------------------
def _declared_symbols():
    names = set()
    for h in ("nbmi.h", "bdmi.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:nbmi|bdmi)_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    import nbmi_native
    lib = ctypes.CDLL(nbmi_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"libnbmi.so lacks {name} declared in include/*.h"
    assert declared == set(nbmi_native.PROTOTYPES), declared ^ set(nbmi_native.PROTOTYPES)
    nbmi_native.load()


def test_backend_enum_and_force_backend():
    from nbody import gpu_backend as gb
    assert {b.value for b in gb.Backend} >= {"hip", "cuda", "metal_barnes_hut", "metal", "cpu"}
    saved = (gb._BACKEND, gb._BACKEND_INFO)
    try:
        gb.force_backend(gb.Backend.CPU)
        assert gb.get_backend() == (gb.Backend.CPU, "Forced: cpu")
        pos = np.zeros((4, 3))
        assert gb.create_gpu_simulation(pos, pos, np.ones(4), 1.0, 0.1, 1.0) is None  # "use CPU" convention
    finally:
        gb._BACKEND, gb._BACKEND_INFO = saved


@pytest.mark.skipif(have_gpu(), reason="checks behaviour on a box without a HIP device")
def test_product_fails_loudly_without_gpu():
    import nbmi_native
    from boids import Flock
    from nbody import NBodySimulation
    from nbody import gpu_backend as gb
    assert nbmi_native.device_count() == 0
    assert gb.detect_backend()[0] == gb.Backend.CPU
    pos = np.zeros((4, 3))
    with pytest.raises(RuntimeError):
        gb.HIPBarnesHutSimulation(pos, pos, np.ones(4), 1.0, 0.1, 1.0, 0.5)
    with pytest.raises(RuntimeError):
        gb.HIPDirectSimulation(pos, pos, np.ones(4), 1.0, 0.1, 1.0)
    saved = (gb._BACKEND, gb._BACKEND_INFO)
    try:
        gb._BACKEND = None
        with pytest.raises(RuntimeError):
            NBodySimulation(16)
    finally:
        gb._BACKEND, gb._BACKEND_INFO = saved
    with pytest.raises(RuntimeError):
        Flock(16)


def test_product_never_imports_oracle():
    """The shipped package must not reference oracle/ (checker only)."""
    import conftest
    pkg = conftest.PKG.PACKAGE_DIR
    for dp, _dn, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dp, fn)).read()
                assert "pyref" not in text and "nbref" not in text and "bdref" not in text, os.path.join(dp, fn)


def test_philox_known_answers():
    """The generator behind the device-side ICs is Philox4x32-10 (Salmon et al. 2011).  Host entry
    point of the same inline function the kernels call; vectors from the Random123 distribution's
    known-answer file, plus an independent Python restatement on random inputs."""
    import ctypes as C
    import nbmi_native
    lib = nbmi_native.load()

    def dev(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        lib.nbmi_philox4x32_10(c, k, o)
        return list(o)

    def ref(ctr, key):
        c, k = list(ctr), list(key)
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
            k = [(k[0] + 0x9E3779B9) & 0xFFFFFFFF, (k[1] + 0xBB67AE85) & 0xFFFFFFFF]
        return c

    assert dev([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert dev([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert dev([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    rng = np.random.RandomState(0)
    for _ in range(50):
        ctr = [int(v) for v in rng.randint(0, 2 ** 32, 4, dtype=np.uint64)]
        key = [int(v) for v in rng.randint(0, 2 ** 32, 2, dtype=np.uint64)]
        assert dev(ctr, key) == ref(ctr, key)


def test_frames_reach_their_name_only_complete(tmp_path):
    """tools.record._atomically: an interrupt inside the write leaves neither the file nor a partial one."""
    from tools import record as rec
    rec.save_frame(tmp_path, 0, np.zeros((5, 3)), np.ones((5, 3)))
    assert rec.get_completed_frames(tmp_path) == 1

    def dies(f):
        f.write(b"half a frame")
        raise KeyboardInterrupt

    with pytest.raises(KeyboardInterrupt):
        rec._atomically(tmp_path / "frame_0001.npz", dies)
    assert rec.get_completed_frames(tmp_path) == 1
    assert sorted(q.name for q in tmp_path.iterdir()) == ["frame_0000.npz"]


def test_oracle_cache_manifest_covers_every_case_and_matches_the_files():
    """tests/oracle_cases.py: every long-trajectory case has its hashes committed in tests/golden/MANIFEST.json, and
    whatever is under tests/cache/ here is what those hashes say (a stale or regenerated file must not reach the GPU
    suite unnoticed; the suite itself asserts the same on load)."""
    import oracle_cases as oc
    man = oc.manifest()
    for case, c in oc.CASES.items():
        assert case in man, f"{case}: no entry in MANIFEST.json (scripts/oracle_cache.py {case})"
        names = {os.path.basename(oc.cache_file(case, k)) for k in c["keep"]}
        assert names == set(man[case]["files"]), (case, sorted(names ^ set(man[case]["files"])))
        assert man[case]["case"]["n"] == c["n"] and man[case]["case"]["dt"] == c["dt"] and man[case]["case"]["seed"] == c["seed"]
        for name, want in man[case]["files"].items():
            f = os.path.join(oc.CACHE, name)
            if os.path.exists(f):
                assert oc.sha256_file(f) == want, name
