"""CPU: libdynam3d_hip.so loads without a GPU and exports every symbol include/dynam3d_hip.h declares."""
import ctypes
import os
import re

from dynam3d_amd import _lib
from dynam3d_amd.build import build_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dynam3d_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_all_declared_symbols():
    build_hip(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_signature_table_covers_kernel_entry_points():
    syms = set(declared_symbols())
    from dynam3d_amd import f32_ops, hip_dense, render, segm, tcnn, train_ops  # noqa: F401  (register the dense / float32 / renderer / segmenter / training signatures)
    bound = set(_lib.SIGNATURES) | set(_lib.FFSTATE_SYMBOLS) | set(_lib.MISC_SYMBOLS) | {"d3d_ff_set_tomb_cell"}
    assert syms <= bound | {"d3d_ff_set_tomb_cell"}, sorted(syms - bound)


def test_loader_binds():
    lib = _lib.load()
    assert lib.d3d_version() >= 100
