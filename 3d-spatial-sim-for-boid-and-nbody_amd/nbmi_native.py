"""ctypes binding of libnbmi.so (C ABI declared in include/nbmi.h and include/bdmi.h).

There is no CPU fallback: if the shared library is missing this module raises at load time,
and if no HIP device is present the constructors raise RuntimeError.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBMI_LIB") or os.path.join(_HERE, "libnbmi.so")  # NBMI_LIB: A/B builds (measurement)

_lib = None

_i64 = C.c_int64
_dbl = C.c_double
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/nbmi.h and include/bdmi.h
PROTOTYPES = {
    "nbmi_device_count": (C.c_int, []),
    "nbmi_last_error": (C.c_char_p, []),
    "nbmi_create": (_vp, [_i64, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, C.c_int, C.c_int]),
    "nbmi_create_generated": (_vp, [C.c_int, _i64, _dbl, C.c_uint64, _dbl, _dbl, _dbl, _dbl, C.c_int, C.c_int]),
    "nbmi_philox4x32_10": (None, [_vp, _vp, _vp]),
    "nbmi_get_masses_f64": (C.c_int, [_vp, _vp]),
    "nbmi_destroy": (None, [_vp]),
    "nbmi_step": (C.c_int, [_vp, _dbl, C.c_int]),
    "nbmi_step_count": (_i64, [_vp]),
    "nbmi_compute_colors": (C.c_int, [_vp, _dbl]),
    "nbmi_get_positions_f32": (C.c_int, [_vp, _vp]),
    "nbmi_get_velocities_f64": (C.c_int, [_vp, _vp]),
    "nbmi_get_colors_f32": (C.c_int, [_vp, _vp]),
    "nbmi_sync": (C.c_int, [_vp]),
    "nbmi_get_positions_f64": (C.c_int, [_vp, _vp]),
    "nbmi_set_state": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_build_tree": (C.c_int, [_vp]),
    "nbmi_get_accelerations_f64": (C.c_int, [_vp, _vp]),
    "nbmi_tree_stats": (C.c_int, [_vp, _vp, _vp, _vp]),
    "nbmi_get_keys": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_get_sort_keys": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_get_cells": (C.c_int, [_vp, _vp, _vp, _i64]),
    "nbmi_enable_timers": (C.c_int, [_vp, C.c_int]),
    "nbmi_get_timers": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "nbmi_walk_counters": (C.c_int, [_vp, _vp]),
    "nbmi_set_shard": (C.c_int, [_vp, _i64, _i64]),
    "nbmi_export_shard": (C.c_int, [_vp, _vp]),
    "nbmi_import_ranks": (C.c_int, [_vp, _vp, _i64, _i64]),
    "nbmi_get_order": (C.c_int, [_vp, _vp]),
    "nbmi_create_owner": (_vp, [_i64, _vp, _vp, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _dbl, _dbl, _dbl, _dbl, C.c_int]),
    "nbmi_owner_count": (_i64, [_vp]),
    "nbmi_owner_boxes_per_rank": (C.c_int, []),
    "nbmi_owner_let_row_bytes": (C.c_int, []),
    "nbmi_owner_set_dt": (C.c_int, [_vp, _dbl]),
    "nbmi_owner_get_ids": (C.c_int, [_vp, _vp]),
    "nbmi_owner_maxabs": (C.c_int, [_vp, _vp]),
    "nbmi_owner_sample": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "nbmi_owner_partition": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "nbmi_owner_adopt": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "nbmi_owner_chain_doubles": (C.c_int, []),
    "nbmi_owner_export_let": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "nbmi_owner_step": (C.c_int, [_vp, _vp, _vp, _dbl]),
    "nbmi_owner_step_facts": (C.c_int, [_vp, _vp]),
    "nbmi_owner_set_all64": (C.c_int, [_vp, C.c_int]),
    "nbmi_visible_points": (C.c_int, [_vp, _vp, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp]),
    "nbmi_set_exchange_sync": (C.c_int, [_vp, C.c_int]),
    "nbmi_set_force_precision": (C.c_int, [_vp, C.c_int, _dbl]),
    "nbmi_force_precision_share": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_stream": (_vp, [_vp]),
    "nbmi_frame_keyframe": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_frame_delta_i16": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_frame_set_previous": (C.c_int, [_vp, _vp, _vp]),
    "nbmi_debug_sort_pairs": (C.c_int, [C.c_int, _i64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "bdmi_create": (_vp, [_i64, _vp, _vp, _vp, _vp, C.c_int]),
    "bdmi_create_slab": (_vp, [_i64, _vp, _vp, _vp, _vp, _i64, _vp, _dbl, _dbl, C.c_int, C.c_int, C.c_int]),
    "bdmi_slab_count": (_i64, [_vp]),
    "bdmi_slab_export": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bdmi_slab_import": (C.c_int, [_vp, _vp, _i64]),
    "bdmi_slab_get": (C.c_int, [_vp, _vp, _i64, _vp]),
    "bdmi_destroy": (None, [_vp]),
    "bdmi_last_error": (C.c_char_p, []),
    "bdmi_step": (C.c_int, [_vp, _dbl, C.c_int]),
    "bdmi_sync": (C.c_int, [_vp]),
    "bdmi_get_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "bdmi_set_state": (C.c_int, [_vp, _vp, _vp, _vp]),
    "bdmi_get_cell_indices": (C.c_int, [_vp, _vp]),
    "bdmi_get_forces": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bdmi_grid_info": (C.c_int, [_vp, _vp, _vp, _vp]),
    "bdmi_enable_timers": (C.c_int, [_vp, C.c_int]),
    "bdmi_get_timers": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "bdmi_visible_vertices": (C.c_int, [_vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _i64, _vp]),
}


def _torch_first():
    """PyTorch-ROCm wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so +
    libhsa-runtime64.so.1, same SONAME as /opt/rocm's newer one).  The two coexist in one
    process only if torch's runtime comes up FIRST; if libnbmi.so initialises the system runtime
    first, a later torch.cuda init reports "No HIP GPUs are available".  So: when torch is
    already imported, make it initialise before libnbmi.so is mapped.  Programs that need both
    (bench.py, nbody/sharded.py) import torch first; programs that never import torch are
    unaffected."""
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            torch.cuda.is_available()
        except Exception:
            pass


def load():
    """Load libnbmi.so (once).  Raises ImportError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    _torch_first()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C <package>/csrc`. "
            "There is no CPU fallback in this package.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().nbmi_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def device_count():
    return int(load().nbmi_device_count())


def ptr(a):
    """Data pointer of a C-contiguous NumPy array (or None)."""
    return None if a is None else a.ctypes.data
