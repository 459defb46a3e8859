"""The CPU oracle (oracle/bdref.c) against goldens from the reference's boids.Flock."""
import numpy as np
import pytest

from conftest import golden


def _stepper(oracle, g):
    params = oracle.boids_params(bounds=float(g["bounds"]))
    st = oracle.FlockStepper(g["pos_0"], g["vel_0"], g["col_0"], params, use_numpy_argsort=True)
    assert st.dim == int(g["grid_dim"]) and st.cell == float(g["cell_size"]) and st.offset == float(g["grid_offset"])
    return st


@pytest.mark.parametrize("tag", ["sparse", "dense", "walls"])
def test_flock_bit_exact(oracle, tag):
    g = golden("boids_" + tag)
    st = _stepper(oracle, g)
    steps = int(g["steps"])
    dt = float(g["dt"])
    for s in range(1, steps + 1):
        st.step(dt)
        if s == 1:
            assert np.array_equal(st.cell_indices, g["cell_indices_1"])
            nz = np.flatnonzero(st.cell_counts)
            assert np.array_equal(nz, g["cell_counts_1_nonzero"])
            assert np.array_equal(st.cell_counts[nz], g["cell_counts_1_values"])
            assert np.array_equal(st.sep, g["sep_1"])
            assert np.array_equal(st.ali, g["ali_1"])
            assert np.array_equal(st.coh, g["coh_1"])
            assert np.array_equal(st.avg, g["avg_1"])
        if s in (1, steps):
            # The golden ran under CPython, where `v ** 2` on a NumPy scalar (flock.py:291-295) is
            # libm pow(), which is not always the correctly rounded v*v that Numba (and this
            # oracle) computes: rare 1-ulp differences in the speed clamp.  Hence ulp tolerance.
            for mine, ref in ((st.pos, g[f"pos_{s}"]), (st.vel, g[f"vel_{s}"]), (st.col, g[f"col_{s}"])):
                assert np.allclose(mine, ref, rtol=4e-16, atol=1e-15), s
                assert (mine != ref).mean() < 1e-3, s


def test_counting_sort_variant_close(oracle):
    """The stable counting sort is another valid within-cell order: same result up to FP sum order."""
    g = golden("boids_dense")
    a = _stepper(oracle, g)
    b = oracle.FlockStepper(g["pos_0"], g["vel_0"], g["col_0"], oracle.boids_params(bounds=float(g["bounds"])),
                            use_numpy_argsort=False)
    a.step(float(g["dt"]))
    b.step(float(g["dt"]))
    assert np.allclose(a.pos, b.pos, rtol=0, atol=1e-11)
    assert np.allclose(a.vel, b.vel, rtol=0, atol=1e-10)


def test_visibility_and_cone_vertices_match_reference(oracle):
    """compute_visibility_numba + build_vertices_numba (flock.py:311-447): masks exact; the float32
    vertices exact too (float64 arithmetic in source order, one rounding on store)."""
    g = golden("visibility_boids")
    for k in range(4):
        cam = g[f"cam_{k}"]
        th, tv = g[f"tan_{k}"]
        mask = oracle.compute_visibility_boids(g["pos"], cam[0:3], cam[3:6], cam[6:9], cam[9:12], float(th), float(tv),
                                               float(g["fog_end"]))
        assert np.array_equal(mask, g[f"mask_{k}"])
        idx = np.where(mask)[0].astype(np.int32)
        verts, vcols = oracle.build_vertices(g["pos"], g["vel"], g["col"], idx, float(g["cone_length"]),
                                             float(g["cone_radius"]))
        assert np.array_equal(verts, g[f"vertices_{k}"])
        assert np.array_equal(vcols, g[f"vert_colors_{k}"])
    # the degenerate boids (zero speed, velocity along world-up) are among the visible ones of pose 0
    assert g["mask_0"][5] and g["mask_0"][6]
