"""The hand-written gfx950 radix sort (csrc/radix.hip) against rocPRIM and NumPy: keys and values must come out
bit-identical (a stable sort has exactly one result).  It is the "device radix sort" of the Morton-key octree
build and replaces np.argsort of the boids grid (reference boids/flock.py:618)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


_CHECK = {}


def _rocprim():
    """tests/native/librocprim_check.so - rocPRIM as the cross-check, in a library of the tests' own (the product
    library does not contain it); built by __graft_entry__.build(), or here if it is missing."""
    if "lib" not in _CHECK:
        import os
        import subprocess
        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
        so = os.path.join(here, "librocprim_check.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", here], check=True)
        lib = C.CDLL(so)
        lib.rocprim_check_sort_pairs.restype = C.c_int
        lib.rocprim_check_sort_pairs.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_int, C.c_void_p]
        _CHECK["lib"] = lib
    return _CHECK["lib"]


def _sort(nat, keys, vals, bits, impl, repeats=1):
    """impl 0: the product's radix sort (through the C ABI's debug hook); impl 1: rocPRIM (test-only library)."""
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    ms = C.c_double(0.0)
    if impl == 0:
        lib = nat.load()
        nat.check(lib.nbmi_debug_sort_pairs(keys.dtype.itemsize, len(keys), nat.ptr(keys), nat.ptr(vals), nat.ptr(ko),
                                            nat.ptr(vo), bits, 0, repeats, C.addressof(ms)), "nbmi_debug_sort_pairs")
    else:
        rc = _rocprim().rocprim_check_sort_pairs(keys.dtype.itemsize, len(keys), nat.ptr(keys), nat.ptr(vals), nat.ptr(ko),
                                                 nat.ptr(vo), bits, repeats, C.addressof(ms))
        assert rc == 0, f"rocprim_check_sort_pairs: {rc}"
    return ko, vo, ms.value


def _cases(rng, n, dtype, bits):
    top = (1 << bits) - 1
    yield "uniform", rng.integers(0, top, n, dtype=np.uint64, endpoint=True).astype(dtype)
    yield "few distinct", (rng.integers(0, 5, n, dtype=np.uint64) * np.uint64(top // 7)).astype(dtype)
    yield "all equal", np.full(n, top // 3, dtype=dtype)
    yield "sorted", np.sort(rng.integers(0, top, n, dtype=np.uint64, endpoint=True)).astype(dtype)
    yield "reversed", np.sort(rng.integers(0, top, n, dtype=np.uint64, endpoint=True))[::-1].astype(dtype).copy()
    # clustered like octant keys of a concentrated system: most pairs share their upper digits
    yield "shared upper digits", ((np.uint64(top) >> np.uint64(2)) ^ rng.integers(0, 1 << min(bits, 20), n, dtype=np.uint64)).astype(dtype)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 12_289, 100_003])
def test_radix_sort_matches_numpy_and_rocprim(gpu, n):
    import nbmi_native as nat
    rng = np.random.default_rng(n)
    for dtype, bits in ((np.uint64, 63), (np.uint64, 17), (np.uint32, 24), (np.uint32, 32)):
        for name, keys in _cases(rng, n, dtype, bits):
            vals = rng.permutation(n).astype(np.uint32)
            ko, vo, _ = _sort(nat, keys, vals, bits, 0)
            order = np.argsort(keys, kind="stable")
            assert np.array_equal(ko, keys[order]), (name, dtype, bits)
            assert np.array_equal(vo, vals[order]), (name, dtype, bits)
            kr, vr, _ = _sort(nat, keys, vals, bits, 1)
            assert np.array_equal(ko, kr) and np.array_equal(vo, vr), (name, dtype, bits)


@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
def test_radix_sort_at_bench_sizes_and_timing(gpu, n):
    """The octree build's own input: the 63-bit octant keys of the bench ICs, and the boids' 24-bit cell indices."""
    import nbmi_native as nat
    from oracle import pyref
    from tools.presets import generate_distribution
    np.random.seed(42)
    p, _, _ = generate_distribution("galaxy" if n == 1_000_000 else "collision", n, 800.0 if n == 1_000_000 else 2000.0, 0.07)
    hi, _ = pyref.body_keys(p, pyref.compute_bounds(p))
    vals = np.arange(n, dtype=np.uint32)
    ko, vo, ms_own = _sort(nat, hi, vals, 63, 0, repeats=10)
    kr, vr, ms_lib = _sort(nat, hi, vals, 63, 1, repeats=10)
    assert np.array_equal(ko, kr) and np.array_equal(vo, vr)
    assert np.all(ko[1:] >= ko[:-1])
    print(f"n={n}: 63-bit keys, own radix sort {ms_own:.3f} ms, rocPRIM {ms_lib:.3f} ms")
    cells = np.random.default_rng(1).integers(0, 8_242_408, n, dtype=np.uint32)
    ko, vo, ms_own = _sort(nat, cells, vals, 24, 0, repeats=10)
    kr, vr, ms_lib = _sort(nat, cells, vals, 24, 1, repeats=10)
    assert np.array_equal(ko, kr) and np.array_equal(vo, vr)
    print(f"n={n}: 24-bit cell indices, own radix sort {ms_own:.3f} ms, rocPRIM {ms_lib:.3f} ms")
