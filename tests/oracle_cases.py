"""Long oracle trajectories the GPU suite compares against, and where they are kept.

TEST INFRASTRUCTURE.  A case is (initial conditions, constants, step size); its strict-IEEE oracle trajectory
(oracle/nbref.c, bit-identical on any x86-64 host) is written once in the build container by
scripts/oracle_cache.py into tests/cache/ (git-ignored, travels with the tree snapshot), and the SHA-256 of every such
file is committed in tests/golden/MANIFEST.json.  `load(case, steps, oracle)` returns the initial conditions and the
oracle positions: from the cache when the file is there AND matches the manifest, otherwise computed on the spot
(minutes of host CPU at 1 M bodies) or - for the 10 M cases that take hours - skipped with the reason.  Never silent:
a file whose hash is not the committed one fails the test.
"""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "tests", "cache")
MANIFEST = os.path.join(ROOT, "tests", "golden", "MANIFEST.json")

# name -> what defines the trajectory.  `every`: only bodies 0, every, 2*every, ... are kept.
CASES = {
    # BASELINE config 2: tools/presets.py:1516-1532.  The input the "auto" force precision was tuned on.
    "galaxy_1m": dict(dist="galaxy", n=1_000_000, seed=42, R=800.0, G=0.07, eps=1.5, theta=0.5, dt=0.05,
                      keep=(10, 20, 30, 50, 100), every=1, stem="oracle_galaxy_1000000"),
    # held-out inputs: never used to choose tau
    "galaxy_1m_seed7": dict(dist="galaxy", n=1_000_000, seed=7, R=800.0, G=0.07, eps=1.5, theta=0.5, dt=0.05,
                            keep=(20, 50, 100), every=1, stem="oracle_galaxy_seed7_1000000"),
    # config 4's constants (tools/presets.py:2424-2440) at a tenth of its size
    "collision_1m": dict(dist="collision", n=1_000_000, seed=43, R=2000.0, G=0.08, eps=6.0, theta=0.5, dt=0.25,
                         keep=(20, 50, 100), every=1, stem="oracle_collision_seed43_1000000"),
    # Plummer sphere through the BH path, accurate_cluster constants (tools/presets.py:1868-1884)
    "cluster_1m": dict(dist="cluster", n=1_000_000, seed=44, R=300.0, G=0.05, eps=1.0, theta=0.5, dt=0.02,
                       keep=(20, 50, 100), every=1, stem="oracle_cluster_seed44_1000000"),
    # second held-out set [r4]: made AFTER the system-wide threshold of "auto" was lowered in response to collision_1m
    # (3.2 x inside the bound with the round-3 thresholds) - inputs nothing was adjusted on
    "collision_1m_b": dict(dist="collision", n=1_000_000, seed=45, R=1200.0, G=0.08, eps=4.0, theta=0.5, dt=0.2,
                           keep=(50, 100), every=1, stem="oracle_collision_seed45_1000000"),
    "galaxy_1m_dt01": dict(dist="galaxy", n=1_000_000, seed=9, R=800.0, G=0.07, eps=1.5, theta=0.5, dt=0.1,
                           keep=(50, 100), every=1, stem="oracle_galaxy_seed9_dt01_1000000"),
    # the reference's LIVE configuration (config/nbody.py:16-17, 57-73: 150 000 bodies, theta 0.8, G 0.1, eps 2.0,
    # R 500; NBodySimulation.update caps dt at 0.02, simulation.py:802)
    "live_150k": dict(dist="galaxy", n=150_000, seed=3, R=500.0, G=0.1, eps=2.0, theta=0.8, dt=0.02,
                      keep=(100,), every=1, stem="oracle_live_galaxy_seed3_150000"),
    # BASELINE config 4 / north_star size: uncapped oracle, every 16th body (100 s of 8 cores per step)
    "collision_10m": dict(dist="collision", n=10_000_000, seed=42, R=2000.0, G=0.08, eps=6.0, theta=0.5, dt=0.25,
                          keep=(10, 20, 50, 100), every=16, stem="oracle_collision_10000000"),
}


def cache_file(case, step):
    c = CASES[case]
    tail = f"_every{c['every']}" if c["every"] > 1 else ""
    return os.path.join(CACHE, f"{c['stem']}_step{step}{tail}.npy")


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def manifest():
    """The "oracle_cache" object of tests/golden/MANIFEST.json (the rest of that file: sizes of the golden fixtures,
    written by oracle/gen_golden.py, which preserves this object)."""
    if not os.path.exists(MANIFEST):
        return {}
    with open(MANIFEST) as f:
        return json.load(f).get("oracle_cache", {})


def write_manifest(cache_entries):
    whole = {}
    if os.path.exists(MANIFEST):
        with open(MANIFEST) as f:
            whole = json.load(f)
    whole["oracle_cache"] = cache_entries
    with open(MANIFEST, "w") as f:
        json.dump(whole, f, indent=1, sort_keys=True)


def initial_conditions(case):
    from tools.presets import generate_distribution
    c = CASES[case]
    np.random.seed(c["seed"])
    return generate_distribution(c["dist"], c["n"], c["R"], c["G"])


def ic_digest(p, v, m):
    """What the cached trajectory belongs to: changes when the generator or its constants change."""
    h = hashlib.sha256()
    for a in (p, v, m):
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def stepper(case, oracle, p, v, m):
    c = CASES[case]
    return oracle.BHStepper(p, v, m, c["theta"], c["G"], c["eps"], 1.0, cap=oracle.UNCAPPED,
                            rows=4 * c["n"] + 4096, fast=False)


def load(case, steps, oracle=None, compute_if_missing=True):
    """-> (p, v, m, {step: oracle positions (every-th rows)}).  Hash-checked against the manifest."""
    import pytest
    c = CASES[case]
    p, v, m = initial_conditions(case)
    man = manifest()
    entry = man.get(case, {})
    files = {k: cache_file(case, k) for k in steps}
    if all(os.path.exists(f) for f in files.values()):
        assert entry, f"tests/cache holds {case} but tests/golden/MANIFEST.json has no entry for it"
        assert entry["ic_sha256"] == ic_digest(p, v, m), \
            f"{case}: the cached trajectory was made from other initial conditions than the generator gives now"
        out = {}
        for k, f in files.items():
            want = entry["files"].get(os.path.basename(f))
            assert want is not None, f"{os.path.basename(f)} is not in MANIFEST.json"
            got = sha256_file(f)
            assert got == want, f"{os.path.basename(f)}: sha256 {got[:16]}.. is not the committed {want[:16]}.."
            out[k] = np.load(f)
        print(f"  ({case}: oracle trajectory from tests/cache, hashes match MANIFEST.json)")
        return p, v, m, out
    missing = [os.path.basename(f) for f in files.values() if not os.path.exists(f)]
    if not compute_if_missing or c["n"] > 2_000_000 or oracle is None:
        pytest.skip(f"{case}: {missing} not under tests/cache (scripts/oracle_cache.py {case} makes them in the build "
                    f"container; too much host CPU to compute inside the suite)")
    print(f"  ({case}: {missing} not cached - computing the oracle here)")
    L = oracle.lib()
    L.nbref_set_num_threads(min(32, int(L.nbref_num_threads())))
    ostep = stepper(case, oracle, p, v, m)
    out = {}
    for k in range(1, max(steps) + 1):
        ostep.step(c["dt"])
        if k in steps:
            out[k] = ostep.pos[::c["every"]].copy()
            want = entry.get("files", {}).get(os.path.basename(files[k]))
            if want is not None:  # the committed hash also pins a freshly computed trajectory
                os.makedirs(CACHE, exist_ok=True)
                np.save(files[k], out[k])
                assert sha256_file(files[k]) == want, f"{case} step {k}: computed oracle differs from MANIFEST.json"
    return p, v, m, out
