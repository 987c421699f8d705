"""ctypes binding of the host bookkeeping half of the C ABI (include/dynam3d_hip.h, d3d_ff_*)."""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


def _p(a: np.ndarray, t=C.c_int32):
    return a.ctypes.data_as(C.POINTER(t))


def bind_ffstate(lib: C.CDLL) -> None:
    lib.d3d_last_error.restype = C.c_char_p
    lib.d3d_ff_create.restype = C.c_void_p
    lib.d3d_ff_create.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.d3d_ff_destroy.argtypes = [C.c_void_p]
    lib.d3d_ff_destroy.restype = None
    lib.d3d_ff_set_tomb_cell.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.d3d_ff_reset.argtypes = [C.c_void_p, C.c_int32]
    lib.d3d_ff_pop.argtypes = [C.c_void_p, C.c_int32]
    lib.d3d_ff_batch_size.argtypes = [C.c_void_p]
    lib.d3d_ff_count.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.d3d_ff_count.restype = C.c_int64
    lib.d3d_ff_apply_hits.argtypes = [C.c_void_p, C.c_int32, i32p, C.c_int32, i32p, i32p, i32p, i32p, C.c_int32]
    lib.d3d_ff_begin_view.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, i32p]
    lib.d3d_ff_plan_merge.argtypes = [C.c_void_p, C.c_int32, i32p, C.c_int32, C.c_int32, C.c_int32, f32p, i32p, f32p, i32p,
                                      i32p, i32p, i32p, i32p, i32p, i32p, C.c_int32]
    lib.d3d_ff_plan_zones.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, i32p, i32p, i32p, i32p, C.c_int32, C.c_int32]
    lib.d3d_ff_end_view.argtypes = [C.c_void_p, C.c_int32, i32p]
    lib.d3d_ff_rebuild_tree.argtypes = [C.c_void_p, C.c_int32, i32p]
    lib.d3d_ff_live_ids.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, i32p, i32p, C.c_int32]
    lib.d3d_ff_export_owner.argtypes = [C.c_void_p, C.c_int32, i32p, C.c_int64]
    lib.d3d_ff_export_members.argtypes = [C.c_void_p, C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int64]
    lib.d3d_ff_export_zone_keys.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, C.c_int32]


class D3DError(RuntimeError):
    pass


class FFState:
    """Thin OO view of a `d3d_ff*` handle.  All arrays are host numpy int32/float32."""

    ROWS, SLOTS, LIVE, ZROWS, ZLIVE, OWNED, TREE = range(7)

    def __init__(self, lib: C.CDLL, compat: str = "reference", patches_per_view: int = 576, num_proposals: int = 2):
        bind_ffstate(lib)
        self.lib = lib
        self.P, self.K = patches_per_view, num_proposals
        self.h = lib.d3d_ff_create(1 if compat == "fixed" else 0, patches_per_view, num_proposals)
        if not self.h:
            raise D3DError("d3d_ff_create failed")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.d3d_ff_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise D3DError(f"{self.lib.d3d_last_error().decode()} (code {rc})")
        return rc

    def reset(self, B):
        self._ck(self.lib.d3d_ff_reset(self.h, B))

    def pop(self, e):
        self._ck(self.lib.d3d_ff_pop(self.h, e))

    def set_tomb_cell(self, c):
        self._ck(self.lib.d3d_ff_set_tomb_cell(self.h, int(c[0]), int(c[1]), int(c[2])))

    @property
    def batch_size(self):
        return self.lib.d3d_ff_batch_size(self.h)

    def count(self, e, which):
        return int(self.lib.d3d_ff_count(self.h, e, which))

    def apply_hits(self, e, hits: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        hits = np.ascontiguousarray(hits, np.int32)
        cap = max(16, self.count(e, self.SLOTS) + 1, self.count(e, self.ZROWS) + 1)
        di, dz = np.empty(cap, np.int32), np.empty(cap, np.int32)
        ni, nz = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.d3d_ff_apply_hits(self.h, e, _p(hits), len(hits), _p(di), C.byref(ni), _p(dz), C.byref(nz), cap))
        return di[:ni.value].copy(), dz[:nz.value].copy()

    def begin_view(self, e):
        rb, k0, ht = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.d3d_ff_begin_view(self.h, e, C.byref(rb), C.byref(k0), C.byref(ht)))
        return rb.value, k0.value, bool(ht.value)

    def plan_merge(self, e, segm, n_seg, k0, k_max, d2, idx, logits, new_cells):
        segm = np.ascontiguousarray(segm, np.int32)
        d2 = np.ascontiguousarray(d2, np.float32)
        idx = np.ascontiguousarray(idx, np.int32)
        logits = np.ascontiguousarray(logits, np.float32)
        new_cells = np.ascontiguousarray(new_cells, np.int32)
        rows_cap = self.count(e, self.OWNED) + self.P + 16
        seg_slot = np.empty(n_seg, np.int32)
        dirty = np.empty(n_seg + 1, np.int32)
        off = np.empty(n_seg + 2, np.int32)
        rows = np.empty(rows_cap, np.int32)
        keff, nd = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.d3d_ff_plan_merge(self.h, e, _p(segm), n_seg, k0, k_max, _p(d2, C.c_float), _p(idx), _p(logits, C.c_float),
                                            _p(new_cells), C.byref(keff), _p(seg_slot), _p(dirty), C.byref(nd), _p(off), _p(rows), rows_cap))
        n = nd.value
        return keff.value, seg_slot, dirty[:n].copy(), off[:n + 1].copy(), rows[:off[n] if n else 0].copy()

    def plan_zones(self, e, dirty_cells, n_seg):
        dirty_cells = np.ascontiguousarray(dirty_cells, np.int32).reshape(-1, 3)
        zcap, mcap = n_seg + 1, self.count(e, self.SLOTS) + 1
        row, mode = np.empty(zcap, np.int32), np.empty(zcap, np.int32)
        off, mem = np.empty(zcap + 1, np.int32), np.empty(mcap, np.int32)
        nt = C.c_int32(0)
        self._ck(self.lib.d3d_ff_plan_zones(self.h, e, _p(dirty_cells), C.byref(nt), _p(row), _p(mode), _p(off), _p(mem), zcap, mcap))
        n = nt.value
        return row[:n].copy(), mode[:n].copy(), off[:n + 1].copy(), mem[:off[n] if n else 0].copy()

    def end_view(self, e):
        t = C.c_int32(0)
        self._ck(self.lib.d3d_ff_end_view(self.h, e, C.byref(t)))
        return t.value

    def live_ids(self, e):
        cap = max(self.count(e, self.SLOTS), self.count(e, self.ZROWS)) + 1
        a, b = np.empty(cap, np.int32), np.empty(cap, np.int32)
        na, nb = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.d3d_ff_live_ids(self.h, e, _p(a), C.byref(na), _p(b), C.byref(nb), cap))
        return a[:na.value].copy(), b[:nb.value].copy()

    # ---- debug / test export ------------------------------------------------------------------
    def export(self, e):
        n = self.count(e, self.ROWS)
        owner = np.empty(max(n, 1), np.int32)
        self._ck(self.lib.d3d_ff_export_owner(self.h, e, _p(owner), len(owner)))
        owner = owner[:n]
        out = {"owner": {int(i): int(owner[i]) for i in np.nonzero(owner >= 0)[0]}}
        for which, name in ((0, "members"), (1, "zmembers")):
            nk = self.count(e, self.SLOTS if which == 0 else self.ZROWS) + 1
            ids, off = np.empty(nk, np.int32), np.empty(nk + 1, np.int32)
            flat = np.empty((n + nk) * (1 if which == 0 else 4) + 16, np.int32)
            m = self._ck(self.lib.d3d_ff_export_members(self.h, e, which, _p(ids), _p(off), _p(flat), len(flat)))
            out[name] = {int(ids[i]): flat[off[i]:off[i + 1]].astype(np.int64) for i in range(m)}
        nk = self.count(e, self.ZROWS) + 1
        cells, ids = np.empty((nk, 3), np.int32), np.empty(nk, np.int32)
        m = self._ck(self.lib.d3d_ff_export_zone_keys(self.h, e, _p(cells), _p(ids), nk))
        out["zkey_cells"] = {tuple(int(c) for c in cells[i]): int(ids[i]) for i in range(m)}
        return out
