"""Device-resident planner of the memory update: the Python side of csrc/ff_plan.h / ff_plan_kernels.hip (d3d_ffdev_*).

`DevicePlanner` owns the int32 state arrays (one row per storage slot, on the feature field's device) and the few counters the HOST
can know without asking the device (rows grow by P per view; the exact slot / zone / edge counts arrive with each view's report and
bound the next view's growth).  `FFDevOps` are the tensor-level wrappers of the seven entry points; `ops.HipOps` inherits them (HIP
kernels on the current stream) and so does the tests' CPU emulation (the same source compiled over host arrays, tests only).

Reference: the dict / list bookkeeping of VLN-FF:362-393, 433-475, 623-691, 694-756, 825/844 -- see csrc/ff_plan.h for how each
dictionary became an array."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from . import _lib
from ._lib import i32, i64, vp

HDR_WORDS = 16
REPORT_WORDS = 16
H_NROWS, H_NOWNED, H_NSLOTS, H_NLIVE, H_NZROWS, H_NZLIVE, H_NZIDS, H_STAMP, H_HAS_TREE, H_TREE_SLOTS, H_NEDGES, H_EDGE_SEL, H_ERR = range(13)
V_NDIRTY, V_DIRTY_ROWS, V_KEFF, V_NTOUCHED, V_ZONE_MEMBERS, V_NSLOTS, V_NZROWS, V_NZIDS, V_NEDGES, V_ERR, V_NLIVE, V_NZLIVE, V_NOWNED = range(13)
ERRORS = {1: "a merge proposal is not a live instance (KeyError in the reference)", 2: "member rows of the merged instances overflow / disagree with the member count",
          4: "zone-member edge table overflow", 8: "instance / zone slot capacity exceeded"}


class FFDevState(C.Structure):
    """d3d_ffdev_state (include/dynam3d_hip.h)."""
    _fields_ = [("hdr", vp), ("rows", vp), ("inst", vp), ("zone", vp), ("edges", vp), ("scratch", vp),
                ("R", i32), ("M", i32), ("Z", i32), ("E", i32), ("W", i32), ("compat_fixed", i32), ("P", i32), ("K", i32), ("tomb", i32 * 3)]


_SIGS = {
    "d3d_ffdev_begin_view": [vp, vp, i32, vp, vp, vp],
    "d3d_ffdev_apply_hits": [vp, vp, i32, vp, i64, vp, vp, vp, i64, vp, vp, i64, i32, vp],
    "d3d_ffdev_plan_merge": [vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp],
    "d3d_ffdev_flatten_merge": [i32, i32, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp],
    "d3d_ffdev_plan_zones": [vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, i64, vp, vp],
    "d3d_ffdev_flatten_zones": [i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp],
    "d3d_ffdev_live_ids": [vp, vp, i32, vp, vp, vp, vp, i32, vp],
}
for _n, _a in _SIGS.items():
    _lib.register(_n, _a)


def bind_ffdev(lib: C.CDLL) -> None:
    for n, a in _SIGS.items():
        fn = getattr(lib, n)
        fn.argtypes, fn.restype = a, C.c_int32


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FFDevOps:
    """Mixin: expects `self.lib` (a CDLL exporting d3d_ffdev_*), `self._ck(rc)` and optionally `self._stream()`."""

    def _ffdev(self):
        if not getattr(self, "_ffdev_bound", False):
            bind_ffdev(self.lib)
            self._ffdev_bound = True
        return self.lib

    def _ffdev_stream(self):
        s = getattr(self, "_stream", None)
        return s() if s is not None else None

    def ffdev_begin_view(self, st: "DevicePlanner", slot):
        B = slot.shape[0]
        k0 = torch.empty((B,), dtype=torch.int32, device=slot.device)
        tree = torch.empty((B,), dtype=torch.int32, device=slot.device)
        self._ck(self._ffdev().d3d_ffdev_begin_view(st.struct(), _p(slot), B, _p(k0), _p(tree), self._ffdev_stream()))
        return k0, tree

    def ffdev_apply_hits(self, st: "DevicePlanner", slot, hits, n_hits, pools):
        B = slot.shape[0]
        self._ck(self._ffdev().d3d_ffdev_apply_hits(st.struct(), _p(slot), B, _p(hits), hits.stride(0) if hits.dim() == 2 and hits.shape[0] else 0, _p(n_hits),
                                                    _p(pools.inst_pos), _p(pools.inst_fts), pools.m_cap, _p(pools.zone_pos), _p(pools.zone_fts), pools.z_cap,
                                                    pools.inst_fts.shape[2], self._ffdev_stream()))

    def ffdev_plan_merge(self, st: "DevicePlanner", slot, order, tok_seg, seg_off, n_seg, n_max, k_max, k0, d2, idx, logits, new_cells, rows_stride, report):
        B, dev = slot.shape[0], slot.device
        seg_slot = torch.empty((B, n_max), dtype=torch.int32, device=dev)
        dirty_inst = torch.empty((B, n_max), dtype=torch.int32, device=dev)
        dirty_off = torch.empty((B, n_max + 1), dtype=torch.int32, device=dev)
        dirty_rows = torch.empty((B, rows_stride), dtype=torch.int32, device=dev)
        self._ck(self._ffdev().d3d_ffdev_plan_merge(st.struct(), _p(slot), B, _p(order), _p(tok_seg), _p(seg_off), _p(n_seg), n_max, k_max, _p(k0), _p(d2), _p(idx),
                                                    _p(logits), _p(new_cells), _p(seg_slot), _p(dirty_inst), _p(dirty_off), _p(dirty_rows), rows_stride, _p(report),
                                                    self._ffdev_stream()))
        return seg_slot, dirty_inst, dirty_off, dirty_rows

    def ffdev_flatten_merge(self, slot, n_max, dirty_inst, dirty_off, dirty_rows, report, grp_off, totals):
        B, dev = slot.shape[0], slot.device
        cap = dirty_rows.numel()
        tok_slot = torch.empty((cap,), dtype=torch.int32, device=dev)
        tok_row = torch.empty((cap,), dtype=torch.int32, device=dev)
        grp_slot = torch.empty((B * n_max,), dtype=torch.int32, device=dev)
        grp_inst = torch.empty((B * n_max,), dtype=torch.int32, device=dev)
        self._ck(self._ffdev().d3d_ffdev_flatten_merge(B, n_max, _p(slot), _p(dirty_inst), _p(dirty_off), _p(dirty_rows), dirty_rows.stride(0), _p(report), _p(tok_slot),
                                                       _p(tok_row), cap, _p(grp_off), _p(grp_slot), _p(grp_inst), _p(totals), self._ffdev_stream()))
        return tok_slot, tok_row, grp_slot, grp_inst

    def ffdev_plan_zones(self, st: "DevicePlanner", slot, dirty_inst, merged_cells, new_cells, n_seg, n_max, mem_stride, report):
        B, dev = slot.shape[0], slot.device
        zone_row = torch.empty((B, n_max), dtype=torch.int32, device=dev)
        zone_mode = torch.empty((B, n_max), dtype=torch.int32, device=dev)
        zone_off = torch.empty((B, n_max + 1), dtype=torch.int32, device=dev)
        zone_mem = torch.empty((B, mem_stride), dtype=torch.int32, device=dev)
        self._ck(self._ffdev().d3d_ffdev_plan_zones(st.struct(), _p(slot), B, _p(dirty_inst), _p(merged_cells), _p(new_cells), _p(n_seg), n_max, _p(zone_row),
                                                    _p(zone_mode), _p(zone_off), _p(zone_mem), mem_stride, _p(report), self._ffdev_stream()))
        return zone_row, zone_mode, zone_off, zone_mem

    def ffdev_flatten_zones(self, slot, n_max, zone_row, zone_mode, zone_off, zone_mem, report, grp_off, totals):
        B, dev = slot.shape[0], slot.device
        cap = zone_mem.numel()
        tok_slot = torch.empty((cap,), dtype=torch.int32, device=dev)
        tok_inst = torch.empty((cap,), dtype=torch.int32, device=dev)
        grp_mode = torch.empty((B * n_max,), dtype=torch.int32, device=dev)
        grp_slot = torch.empty((B * n_max,), dtype=torch.int32, device=dev)
        grp_row = torch.empty((B * n_max,), dtype=torch.int32, device=dev)
        self._ck(self._ffdev().d3d_ffdev_flatten_zones(B, n_max, _p(slot), _p(zone_row), _p(zone_mode), _p(zone_off), _p(zone_mem), zone_mem.stride(0), _p(report),
                                                       _p(tok_slot), _p(tok_inst), cap, _p(grp_off), _p(grp_mode), _p(grp_slot), _p(grp_row), _p(totals),
                                                       self._ffdev_stream()))
        return tok_slot, tok_inst, grp_mode, grp_slot, grp_row

    def ffdev_live_ids(self, st: "DevicePlanner", slot, max_ids: int):
        B, dev = slot.shape[0], slot.device
        inst_ids = torch.empty((B, max_ids), dtype=torch.int32, device=dev)
        zone_ids = torch.empty((B, max_ids), dtype=torch.int32, device=dev)
        n_inst = torch.empty((B,), dtype=torch.int32, device=dev)
        n_zone = torch.empty((B,), dtype=torch.int32, device=dev)
        self._ck(self._ffdev().d3d_ffdev_live_ids(st.struct(), _p(slot), B, _p(inst_ids), _p(n_inst), _p(zone_ids), _p(n_zone), max_ids, self._ffdev_stream()))
        return inst_ids, n_inst, zone_ids, n_zone


class DevicePlanner:
    """State arrays + host-side counters of the device planner.  Same `count(e, which)` codes as `_ffstate.FFState`."""

    ROWS, SLOTS, LIVE, ZROWS, ZLIVE, OWNED, TREE = range(7)

    def __init__(self, compat: str, patches_per_view: int, num_proposals: int, device):
        self.compat_fixed = 1 if compat == "fixed" else 0
        self.P, self.K = int(patches_per_view), int(num_proposals)
        self.device = torch.device(device)
        self.tomb = (0, 0, 0)
        self.hdr = None
        self._struct = None
        self.reset(0, 0, 0, 0)

    # ---- storage ---------------------------------------------------------------------------------------------------------------
    def reset(self, S: int, R: int, M: int, Z: int, tomb=(0, 0, 0)):
        dev = self.device
        self.tomb = tuple(int(t) for t in tomb)
        self.S, self.R, self.M, self.Z = S, R, M, Z
        self.E = max(4096, 2 * M)
        self.W = max(8 * self.P + 16, M + Z)
        z = lambda *s: torch.zeros(s, dtype=torch.int32, device=dev)
        self.hdr = z(S, HDR_WORDS)
        self.rows = torch.full((S, 3, R), -1, dtype=torch.int32, device=dev)
        self.inst = z(S, 6, M)
        self.zone = z(S, 8, Z)
        self.edges = z(S, 2, 2, self.E)
        self.scratch = z(S, self.W)
        self._struct = None
        # what the host knows per ENVIRONMENT (exact after every view's report; rows are known a priori)
        self.n_rows: List[int] = [0] * S
        self.n_slots: List[int] = [0] * S
        self.n_live: List[int] = [0] * S
        self.n_zrows: List[int] = [0] * S
        self.n_zids: List[int] = [0] * S
        self.n_zlive: List[int] = [0] * S
        self.n_edges: List[int] = [0] * S
        self.n_owned: List[int] = [0] * S
        self.stale = False                   # True between a device-side deletion and the next view report (see `count`)
        self.env_slots = None                # environment -> state slot (the feature field's own list, shared: `pop` keeps it in step)

    def pop(self, e: int):
        for a in (self.n_rows, self.n_slots, self.n_live, self.n_zrows, self.n_zids, self.n_zlive, self.n_edges, self.n_owned):
            a.pop(e)

    @property
    def batch_size(self):
        return len(self.n_rows)

    def _regrow(self, name: str, dim: int, new: int, fill: int = 0):
        old = getattr(self, name)
        shape = list(old.shape)
        keep = shape[dim]
        shape[dim] = new
        t = torch.full(shape, fill, dtype=torch.int32, device=old.device)
        t.narrow(dim, 0, keep).copy_(old)
        setattr(self, name, t)
        self._struct = None

    def ensure(self, R: int = 0, M: int = 0, Z: int = 0, E: int = 0):
        """Capacities at least these (arrays double); called by the feature field together with its pools."""
        if R > self.R:
            self.R = max(R, 2 * self.R)
            self._regrow("rows", 2, self.R, -1)
        if M > self.M:
            self.M = max(M, 2 * self.M)
            self._regrow("inst", 2, self.M)
        if Z > self.Z:
            self.Z = max(Z, 2 * self.Z)
            self._regrow("zone", 2, self.Z)
        if E > self.E:
            self.E = max(E, 2 * self.E)
            self._regrow("edges", 3, self.E)
        W = max(8 * self.P + 16, self.M + self.Z)
        if W > self.W:
            self.W = W
            self.scratch = torch.zeros((self.S, W), dtype=torch.int32, device=self.device)
            self._struct = None

    def struct(self):
        if self._struct is None:
            s = FFDevState()
            s.hdr, s.rows, s.inst, s.zone, s.edges, s.scratch = (t.data_ptr() for t in (self.hdr, self.rows, self.inst, self.zone, self.edges, self.scratch))
            s.R, s.M, s.Z, s.E, s.W = self.R, self.M, self.Z, self.E, self.W
            s.compat_fixed, s.P, s.K = self.compat_fixed, self.P, self.K
            s.tomb[0], s.tomb[1], s.tomb[2] = self.tomb
            self._struct = s
        return C.byref(self._struct)

    # ---- host counters ---------------------------------------------------------------------------------------------------------
    def count(self, e: int, which: int) -> int:
        if which == self.TREE:
            raise NotImplementedError("the tree size lives in the device header: planner.header(slot)[H_TREE_SLOTS]")
        if self.stale and which in (self.LIVE, self.ZLIVE, self.OWNED):
            # a frustum deletion (d3d_ffdev_apply_hits) ran since the last view report: it changes these three counters on the device
            # and reports nothing to the host (no device-to-host read on the step's path) -- a query in between reads the header
            slot = self.env_slots[e] if self.env_slots is not None else e
            h = self.header(slot)
            return int(h[{self.LIVE: H_NLIVE, self.ZLIVE: H_NZLIVE, self.OWNED: H_NOWNED}[which]])
        return int((self.n_rows, self.n_slots, self.n_live, self.n_zrows, self.n_zlive, self.n_owned)[which][e])

    def take_report_envs(self, envs, rep: np.ndarray):
        """rep (len(envs), 16): this view's report words.  Raises on a planner error, else refreshes the counters."""
        err = int(np.bitwise_or.reduce(rep[:, V_ERR])) if len(rep) else 0
        if err:
            raise RuntimeError("device planner: " + "; ".join(m for b, m in ERRORS.items() if err & b))
        for j, e in enumerate(envs):
            r = rep[j]
            self.n_slots[e], self.n_live[e], self.n_zrows[e], self.n_zids[e] = int(r[V_NSLOTS]), int(r[V_NLIVE]), int(r[V_NZROWS]), int(r[V_NZIDS])
            self.n_zlive[e], self.n_edges[e], self.n_owned[e] = int(r[V_NZLIVE]), int(r[V_NEDGES]), int(r[V_NOWNED])
        if len(envs) == self.batch_size:
            self.stale = False                                           # every environment's counters are this view's

    def header(self, slot: int) -> np.ndarray:
        return self.hdr[slot].cpu().numpy()

    # ---- debug / test export (the dictionaries of `_ffstate.FFState.export`, rebuilt from the arrays) ----------------------------------
    def export(self, slot: int):
        h = self.header(slot)
        n_rows, ns, nz, ne, sel = int(h[H_NROWS]), int(h[H_NSLOTS]), int(h[H_NZIDS]), int(h[H_NEDGES]), int(h[H_EDGE_SEL])
        rows = self.rows[slot].cpu().numpy()
        owner, stamp_of_pid, pid_of_stamp = rows[0, :n_rows], rows[1, :n_rows], rows[2, :n_rows]
        inst, zone = self.inst[slot].cpu().numpy(), self.zone[slot].cpu().numpy()
        edges = self.edges[slot, sel].cpu().numpy()
        out = {"owner": {int(i): int(owner[i]) for i in np.nonzero(owner >= 0)[0]}}
        members = {}
        u = np.nonzero(pid_of_stamp >= 0)[0]
        u = u[stamp_of_pid[pid_of_stamp[u]] == u]                    # stamps whose patch id was not recycled since
        pids = pid_of_stamp[u]                                       # ... in push order
        own = owner[pids]
        for i in sorted((i for i in range(ns) if inst[0, i]), key=lambda i: inst[1, i]):
            members[int(i)] = pids[own == i].astype(np.int64)
        out["members"] = members
        zm = {}
        for z in sorted((z for z in range(nz) if zone[0, z]), key=lambda z: zone[1, z]):
            zm[int(z)] = np.asarray([int(edges[1, k]) for k in range(ne) if edges[0, k] == z], np.int64)
        out["zmembers"] = zm
        out["zkey_cells"] = {(int(zone[5, z]), int(zone[6, z]), int(zone[7, z])): int(z)
                             for z in sorted((z for z in range(nz) if zone[0, z]), key=lambda z: zone[2, z])}
        return out
