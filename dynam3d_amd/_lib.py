"""Loads libdynam3d_hip.so (the C-ABI product library) and declares the ctypes signatures of
include/dynam3d_hip.h.  There is NO fallback: if the library is missing or does not load, the
product raises -- a CPU path would void every parity claim."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdynam3d_hip.so")

_lib = None

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (every function returns int32 unless listed in _RESTYPE)
SIGNATURES = {
    "d3d_device_info": [vp, vp, vp],
    "d3d_preprocess_depth": [vp, vp, i32, i32, i32, f32, f32, vp],
    "d3d_resize_nearest_preprocess": [vp, vp, i32, i32, i32, i32, i32, f32, f32, vp],
    "d3d_unproject_append": [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, i64, vp],
    "d3d_append_fts": [vp, i32, vp, vp, i32, i32, vp, i64, vp],
    "d3d_patch_3d_info": [vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp],
    "d3d_frustum_cull": [vp, vp, vp, vp, i64, vp, vp, i32, i32, vp, i32, i32, vp, f32, f32, f32, f32, f32, f32, f32, vp, vp, i32, vp, vp],
    "d3d_frustum_mask": [vp, i64, vp, i32, i32, vp, f32, f32, f32, f32, f32, f32, f32, vp, vp],
    "d3d_patch_segm_from_masks": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "d3d_frustum_cull_pinhole": [vp, vp, vp, vp, i64, vp, vp, i32, i32, vp, i32, i32, vp, f32, f32, f32, vp, vp, i32, vp, vp],
    "d3d_frustum_mask_pinhole": [vp, i64, vp, i32, i32, vp, f32, f32, f32, vp, vp],
    "d3d_unproject_pinhole_append": [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i64, vp],
    "d3d_knn": [vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, vp, vp, vp],
    "d3d_knn_radius": [vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, f32, vp, vp, vp],
    "d3d_knn_chunked": [vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "d3d_group_stats7": [vp, vp, vp, i64, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, i64, vp],
    "d3d_group_stats4": [vp, i64, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, i64, vp],
    "d3d_gather_fts": [vp, i64, vp, vp, i32, vp, vp],
    "d3d_gather_rows_f32": [vp, i64, i32, vp, vp, i32, vp, vp],
    "d3d_scatter_rows_f32": [vp, i64, i32, vp, vp, i32, vp, vp, vp],
    "d3d_fill_rows_f32": [vp, i64, i32, vp, vp, i32, f32, vp],
    "d3d_merge_input": [vp, vp, i64, vp, vp, vp, vp, vp, i32, vp, vp],
    "d3d_agent_frame_compact": [vp, vp, i64, vp, vp, vp, i32, i32, vp, f32, vp, vp, vp, vp, vp],
}

# host bookkeeping half (bound in _ffstate.bind_ffstate); listed for the export test
FFSTATE_SYMBOLS = ["d3d_ff_create", "d3d_ff_destroy", "d3d_ff_set_tomb_cell", "d3d_ff_reset", "d3d_ff_pop", "d3d_ff_batch_size", "d3d_ff_count",
                   "d3d_ff_apply_hits", "d3d_ff_begin_view", "d3d_ff_plan_merge", "d3d_ff_plan_zones", "d3d_ff_end_view",
                   "d3d_ff_rebuild_tree", "d3d_ff_live_ids", "d3d_ff_export_owner", "d3d_ff_export_members",
                   "d3d_ff_export_zone_keys"]
MISC_SYMBOLS = ["d3d_last_error", "d3d_version"]


class NativeLibraryMissing(RuntimeError):
    pass


def register(name, argtypes):
    """Other modules (dense/gemm/attention wrappers) add their signatures here before load()."""
    SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int32


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found -- build it with `python -m dynam3d_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the product path.")
    import torch  # noqa: F401  -- FIRST: torch brings its own libamdhip64; a process that loaded this library (and with it /opt/rocm's
    #                  runtime) before torch ends up with two HIP runtimes, and the one behind this library sees no device (hipErrorNoDevice)
    lib = C.CDLL(LIB_PATH)
    lib.d3d_last_error.restype = C.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError == library out of date: fail loudly
        fn.argtypes = argtypes
        fn.restype = C.c_int32
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libdynam3d_hip: {_lib.d3d_last_error().decode()} (code {rc})")


def current_stream_ptr():
    """The current HIP stream of the current device as a ctypes void pointer.  `torch.cuda.current_stream().cuda_stream` builds a Python
    Stream object through four layers of device-index helpers (~8 us); a step makes ~400 launches, and the 3D-token update's launches
    are a latency chain on the host.  The raw getters are two C calls."""
    import torch
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
