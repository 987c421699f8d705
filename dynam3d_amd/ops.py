"""Tensor-level wrappers over the C ABI (include/dynam3d_hip.h).  Every method launches the HIP
kernel on torch's current stream and returns torch tensors; torch is used only for device memory
and streams.  `HipOps()` raises if libdynam3d_hip.so is missing -- no fallback."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .ff_plan import FFDevOps

FTS = 768
POSE_FLOATS = 8


def make_pose(position_habitat: Sequence[float], heading: float) -> np.ndarray:
    """d3d_pose (include/dynam3d_hip.h): python-float trigonometry rounded to float32 exactly like the
    reference's `tensor * math.cos(...)` (VLN-FF:95-99, 290-291, 830-836)."""
    p = np.empty(POSE_FLOATS, np.float32)
    p[0], p[1], p[2] = np.float32(position_habitat[0]), np.float32(-position_habitat[2]), np.float32(position_habitat[1])
    p[3], p[4] = np.float32(math.cos(heading)), np.float32(math.sin(heading))
    p[5], p[6] = np.float32(math.cos(-heading)), np.float32(math.sin(-heading))
    p[7] = np.float32(heading)
    return p


def pinhole_views(intrinsics, extrinsics) -> np.ndarray:
    """d3d_pinhole_view rows: rows 0..2 of each 4x4 world->camera matrix, then K[:3,:3], float32 (PRE-FF:101-108)."""
    out = np.zeros((len(intrinsics), 21), np.float32)
    for i, (K, V) in enumerate(zip(intrinsics, extrinsics)):
        out[i, :12] = np.asarray(V, np.float32)[:3, :4].reshape(-1)
        out[i, 12:] = np.asarray(K, np.float32)[:3, :3].reshape(-1)
    return out


def pinhole_unproject_rows(intrinsics, rots, trans, scale_tan: float, depth_scale: float, depth_trunc: float) -> np.ndarray:
    """d3d_pinhole_unproject rows as 18 float64 words: fx, fy, cx, cy, R[9], T[3], then (scale_tan, depth_scale) and
    (depth_trunc, pad) packed as float32 pairs."""
    out = np.zeros((len(intrinsics), 18), np.float64)
    tail = np.zeros((len(intrinsics), 4), np.float32)
    for i, (K, R, T) in enumerate(zip(intrinsics, rots, trans)):
        K = np.asarray(K, np.float64)
        out[i, :4] = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        out[i, 4:13] = np.asarray(R, np.float64).reshape(-1)
        out[i, 13:16] = np.asarray(T, np.float64).reshape(-1)
        tail[i] = scale_tan, depth_scale, depth_trunc, 0.0
    out[:, 16:18] = tail.view(np.float64)
    return out


@dataclass
class CameraTables:
    """Per-patch tangent tables (VLN-FF:283-287): python-float arithmetic rounded to float32 exactly as
    the reference's list comprehension does; uploaded once per camera setting."""
    H: int
    W: int
    hfov: float
    vfov: float
    tan_xy: torch.Tensor
    tan_z: torch.Tensor
    dir0: torch.Tensor
    th: float

    @staticmethod
    def build(H, W, hfov, vfov, device) -> "CameraTables":
        hW, hH = W // 2, H // 2
        th, tv = math.tan(math.pi * hfov / 360.0), math.tan(math.pi * vfov / 360.0)
        row_xy = np.array([i / hW + 1 / W for i in range(-hW, hW)], np.float32)
        tan_xy = (np.tile(row_xy, H) * np.float32(th)).astype(np.float32)
        col_z = np.array([i / hH - 1 / H for i in range(hH, -hH, -1)], np.float32)
        tan_z = (np.repeat(col_z, W) * np.float32(tv)).astype(np.float32)
        dir0 = (-np.arctan(tan_xy)).astype(np.float32)
        t = lambda a: torch.from_numpy(a).to(device)
        return CameraTables(H, W, hfov, vfov, t(tan_xy), t(tan_z), t(dir0), float(np.float32(th)))


@dataclass
class Pools:
    rows_pos: torch.Tensor    # (S, n_cap, 3) f32
    rows_fts: torch.Tensor    # (S, n_cap, 768) f16
    rows_dir: torch.Tensor    # (S, n_cap) f32
    rows_scale: torch.Tensor  # (S, n_cap) f32
    inst_pos: torch.Tensor    # (S, m_cap, 3) f32
    inst_fts: torch.Tensor    # (S, m_cap, 768) f32
    tree_pos: torch.Tensor    # (S, m_cap, 3) f32  snapshot used by the KNN (kd-tree rebuild points)
    zone_pos: torch.Tensor    # (S, z_cap, 3) f32
    zone_fts: torch.Tensor    # (S, z_cap, 768) f32

    @property
    def n_cap(self):
        return self.rows_pos.shape[1]

    @property
    def m_cap(self):
        return self.inst_pos.shape[1]

    @property
    def z_cap(self):
        return self.zone_pos.shape[1]

    @staticmethod
    def allocate(S, n_cap, m_cap, z_cap, device) -> "Pools":
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=device)
        return Pools(z(S, n_cap, 3), z(S, n_cap, FTS, dt=torch.float16), z(S, n_cap), z(S, n_cap), z(S, m_cap, 3),
                     z(S, m_cap, FTS), z(S, m_cap, 3), z(S, z_cap, 3), z(S, z_cap, FTS))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipOps(FFDevOps):
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()

    # -- helpers ------------------------------------------------------------------------------
    @staticmethod
    def _stream():
        return _lib.current_stream_ptr()

    @staticmethod
    def _ck(rc):
        _lib.check(rc)

    def device_info(self):
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.d3d_device_info(C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- a1 / a2 --------------------------------------------------------------------------------
    def preprocess_depth(self, depth: torch.Tensor, lo=0.0, hi=10.0) -> torch.Tensor:
        d = depth.contiguous().float()
        B, H, W = d.shape[0], d.shape[1], d.shape[2]
        out = torch.empty_like(d)
        self._ck(self.lib.d3d_preprocess_depth(_ptr(d), _ptr(out), B, H, W, lo, hi, self._stream()))
        return out

    def resize_nearest_preprocess(self, depth: torch.Tensor, h: int, w: int, lo=0.0, hi=10.0) -> torch.Tensor:
        d = depth.contiguous().float()
        B, H, W = d.shape[0], d.shape[1], d.shape[2]
        out = torch.empty((B, h, w), dtype=torch.float32, device=d.device)
        self._ck(self.lib.d3d_resize_nearest_preprocess(_ptr(d), _ptr(out), B, H, W, h, w, lo, hi, self._stream()))
        return out

    # -- a5 / a13 -------------------------------------------------------------------------------
    def unproject_append(self, depth24, pose, slot, row_base, cam: CameraTables, pools: Pools):
        n, P = depth24.shape
        self._ck(self.lib.d3d_unproject_append(_ptr(depth24), _ptr(pose), _ptr(slot), _ptr(row_base), n, P, cam.W, _ptr(cam.tan_xy),
                                               _ptr(cam.tan_z), _ptr(cam.dir0), cam.th, _ptr(pools.rows_pos), _ptr(pools.rows_dir),
                                               _ptr(pools.rows_scale), pools.n_cap, self._stream()))

    def append_fts(self, grid, slot, row_base, pools: Pools):
        n, P = grid.shape[0], grid.shape[1]
        assert grid.is_contiguous() and grid.dtype in (torch.float32, torch.float16)
        self._ck(self.lib.d3d_append_fts(_ptr(grid), 1 if grid.dtype == torch.float16 else 0, _ptr(slot), _ptr(row_base), n, P,
                                         _ptr(pools.rows_fts), pools.n_cap, self._stream()))

    def patch_3d_info(self, depth24, cam: CameraTables):
        N, P = depth24.shape
        outs = [torch.empty((N, P), dtype=torch.float32, device=depth24.device) for _ in range(5)]
        self._ck(self.lib.d3d_patch_3d_info(_ptr(depth24), N, P, cam.W, _ptr(cam.tan_xy), _ptr(cam.tan_z), _ptr(cam.dir0), cam.th,
                                            *[_ptr(o) for o in outs], self._stream()))
        return outs

    # -- a4 -------------------------------------------------------------------------------------
    def frustum_cull(self, pools: Pools, slot, n_rows, max_rows, depth, pose, intr, near, far, slack, hits, n_hits, mask=None):
        n, Hd, Wd = depth.shape
        fx, fy, cx, cy = intr
        self._ck(self.lib.d3d_frustum_cull(_ptr(pools.rows_pos), _ptr(pools.rows_fts), _ptr(pools.rows_dir), _ptr(pools.rows_scale),
                                           pools.n_cap, _ptr(slot), _ptr(n_rows), n, max_rows, _ptr(depth), Hd, Wd, _ptr(pose),
                                           fx, fy, cx, cy, near, far, slack, _ptr(hits), _ptr(n_hits), hits.shape[1], _ptr(mask),
                                           self._stream()))

    def frustum_mask(self, points, depth, pose_host: np.ndarray, intr, near, far, slack):
        points, depth = points.contiguous(), depth.contiguous()
        N = points.shape[0]
        Hd, Wd = depth.shape
        mask = torch.empty(N, dtype=torch.uint8, device=points.device)
        ph = np.ascontiguousarray(pose_host, np.float32)
        fx, fy, cx, cy = intr
        self._ck(self.lib.d3d_frustum_mask(_ptr(points), N, _ptr(depth), Hd, Wd, ph.ctypes.data_as(C.c_void_p), fx, fy, cx, cy,
                                           near, far, slack, _ptr(mask), self._stream()))
        return mask

    # -- a6 (after the segmenter network) ---------------------------------------------------------------
    def patch_segm_from_masks(self, masks: torch.Tensor, mask_off: Sequence[int], h: int, w: int):
        """masks (total,H,W) uint8 {0,1} on the device; mask_off (n_img+1) host ints -> (segm (n_img,1,h,w) int64, n_seg (n_img,) int32)."""
        n_img = len(mask_off) - 1
        assert masks.dtype == torch.uint8 and masks.is_contiguous()
        _, H, W = masks.shape
        off = torch.tensor(list(mask_off), dtype=torch.int32, device=masks.device)
        segm = torch.empty((n_img, h * w), dtype=torch.int32, device=masks.device)
        n_seg = torch.empty((n_img,), dtype=torch.int32, device=masks.device)
        mx = max([mask_off[i + 1] - mask_off[i] for i in range(n_img)] + [0])
        self._ck(self.lib.d3d_patch_segm_from_masks(_ptr(masks), _ptr(off), n_img, mx, H, W, h, w, _ptr(segm), _ptr(n_seg), self._stream()))
        return segm.view(n_img, 1, h, w).long(), n_seg

    # -- intrinsics / extrinsics path --------------------------------------------------
    def frustum_cull_pinhole(self, pools: Pools, slot, n_rows, max_rows, depth, views, near, far, slack, hits, n_hits, mask=None):
        """views: (n_env, 21) float32 device tensor = d3d_pinhole_view rows (see `pinhole_views`)."""
        n, Hd, Wd = depth.shape
        self._ck(self.lib.d3d_frustum_cull_pinhole(_ptr(pools.rows_pos), _ptr(pools.rows_fts), _ptr(pools.rows_dir), _ptr(pools.rows_scale),
                                                   pools.n_cap, _ptr(slot), _ptr(n_rows), n, max_rows, _ptr(depth), Hd, Wd, _ptr(views),
                                                   near, far, slack, _ptr(hits), _ptr(n_hits), hits.shape[1], _ptr(mask), self._stream()))

    def frustum_mask_pinhole(self, points, depth, view_host: np.ndarray, near, far, slack):
        points, depth = points.contiguous(), depth.contiguous()
        N = points.shape[0]
        Hd, Wd = depth.shape
        mask = torch.empty(N, dtype=torch.uint8, device=points.device)
        vh = np.ascontiguousarray(view_host, np.float32)
        self._ck(self.lib.d3d_frustum_mask_pinhole(_ptr(points), N, _ptr(depth), Hd, Wd, vh.ctypes.data_as(C.c_void_p), near, far, slack,
                                                   _ptr(mask), self._stream()))
        return mask

    def unproject_pinhole_append(self, depth, cams, slot, row_base, h, w, input_width, pools: Pools):
        """depth (n_env,Hd,Wd) f32 raw sensor units; cams: (n_env, 18) float64 device tensor = d3d_pinhole_unproject rows."""
        n, Hd, Wd = depth.shape
        self._ck(self.lib.d3d_unproject_pinhole_append(_ptr(depth), Hd, Wd, _ptr(cams), _ptr(slot), _ptr(row_base), n, h, w, input_width,
                                                       _ptr(pools.rows_pos), _ptr(pools.rows_dir), _ptr(pools.rows_scale), pools.n_cap,
                                                       self._stream()))

    # -- a8 -------------------------------------------------------------------------------------
    def knn(self, points, point_stride, n_points, queries, query_stride, n_queries, k, n_batch, max_queries, k_max, radius=None):
        """`radius`: only neighbours inside it are needed (the renderer discards the rest): d3d_knn_radius, exact inside the radius."""
        dev = points.device
        d2 = torch.full((n_batch, max_queries, k_max), float("inf"), dtype=torch.float32, device=dev)
        idx = torch.full((n_batch, max_queries, k_max), -1, dtype=torch.int32, device=dev)
        if radius is not None:
            self._ck(self.lib.d3d_knn_radius(_ptr(points), point_stride, _ptr(n_points), _ptr(queries), query_stride, _ptr(n_queries), _ptr(k),
                                             n_batch, max_queries, k_max, float(radius), _ptr(d2), _ptr(idx), self._stream()))
            return d2, idx
        cap = point_stride // 3                                     # capacity of one environment's point set
        n_wg = n_batch * ((max_queries + 255) // 256)
        if cap >= 16384 and n_wg < 512:
            # a large point set under few queries (the Pretrain GT cloud: 4 608 queries x 2e5 points): cut the points into chunks so that the
            # launch fills the chip (d3d_knn_chunked: bit-identical results)
            n_chunks = int(min(256, max(2, min(cap // 2048, 1024 // max(n_wg, 1)))))
            ws_d2 = torch.empty((n_batch, max_queries, n_chunks, k_max), dtype=torch.float32, device=dev)
            ws_idx = torch.empty((n_batch, max_queries, n_chunks, k_max), dtype=torch.int32, device=dev)
            self._ck(self.lib.d3d_knn_chunked(_ptr(points), point_stride, _ptr(n_points), _ptr(queries), query_stride, _ptr(n_queries), _ptr(k),
                                              n_batch, max_queries, k_max, n_chunks, _ptr(ws_d2), _ptr(ws_idx), _ptr(d2), _ptr(idx), self._stream()))
            return d2, idx
        self._ck(self.lib.d3d_knn(_ptr(points), point_stride, _ptr(n_points), _ptr(queries), query_stride, _ptr(n_queries), _ptr(k),
                                  n_batch, max_queries, k_max, _ptr(d2), _ptr(idx), self._stream()))
        return d2, idx

    # -- a7 / a10 / a11 ---------------------------------------------------------------------------
    def group_stats7(self, pools: Pools, tok_slot, tok_row, grp_off, G, cell_len, inst_pos=None, grp_slot=None, grp_inst=None):
        T = tok_row.shape[0]
        dev = tok_row.device
        centroid = torch.empty((G, 3), dtype=torch.float32, device=dev)
        cell = torch.empty((G, 3), dtype=torch.int32, device=dev)
        geom = torch.empty((T, 7), dtype=torch.float32, device=dev)
        self._ck(self.lib.d3d_group_stats7(_ptr(pools.rows_pos), _ptr(pools.rows_dir), _ptr(pools.rows_scale), pools.n_cap,
                                           _ptr(tok_slot), _ptr(tok_row), _ptr(grp_off), G, T, cell_len[0], cell_len[1], cell_len[2],
                                           _ptr(centroid), _ptr(cell), _ptr(geom), _ptr(inst_pos), _ptr(grp_slot), _ptr(grp_inst),
                                           pools.m_cap, self._stream()))
        return centroid, cell, geom

    def group_stats4(self, pools: Pools, tok_slot, tok_inst, grp_off, grp_mode, grp_slot, grp_zone_row, G, cell_len):
        T = tok_inst.shape[0]
        geom = torch.empty((T, 4), dtype=torch.float32, device=tok_inst.device)
        self._ck(self.lib.d3d_group_stats4(_ptr(pools.inst_pos), pools.m_cap, _ptr(tok_slot), _ptr(tok_inst), _ptr(grp_off), _ptr(grp_mode),
                                           _ptr(grp_slot), _ptr(grp_zone_row), G, T, cell_len[0], cell_len[1], cell_len[2], _ptr(geom),
                                           _ptr(pools.zone_pos), pools.z_cap, self._stream()))
        return geom

    # -- row movers -------------------------------------------------------------------------------
    def gather_fts(self, pools: Pools, tok_slot, tok_row):
        T = tok_row.shape[0]
        out = torch.empty((T, FTS), dtype=torch.float32, device=tok_row.device)
        self._ck(self.lib.d3d_gather_fts(_ptr(pools.rows_fts), pools.n_cap, _ptr(tok_slot), _ptr(tok_row), T, _ptr(out), self._stream()))
        return out

    def gather_rows(self, pool, slot, row):
        T, D = row.shape[0], pool.shape[2]
        out = torch.empty((T, D), dtype=torch.float32, device=row.device)
        self._ck(self.lib.d3d_gather_rows_f32(_ptr(pool), pool.shape[1], D, _ptr(slot), _ptr(row), T, _ptr(out), self._stream()))
        return out

    def scatter_rows(self, pool, slot, row, src, src_row=None):
        T, D = row.shape[0], pool.shape[2]
        assert src.is_contiguous() and src.dtype == torch.float32 and src.shape[-1] == D
        self._ck(self.lib.d3d_scatter_rows_f32(_ptr(pool), pool.shape[1], D, _ptr(slot), _ptr(row), T, _ptr(src), _ptr(src_row), self._stream()))

    def fill_rows(self, pool, slot, row, value: float):
        T, D = row.shape[0], pool.shape[2]
        self._ck(self.lib.d3d_fill_rows_f32(_ptr(pool), pool.shape[1], D, _ptr(slot), _ptr(row), T, value, self._stream()))

    # -- a9 / a12 ---------------------------------------------------------------------------------
    def merge_input(self, pools: Pools, new_fts, new_pos, pair_slot, pair_inst, pair_new):
        R = pair_new.shape[0]
        out = torch.empty((R, 2 * FTS + 3), dtype=torch.float32, device=new_fts.device)
        self._ck(self.lib.d3d_merge_input(_ptr(pools.inst_fts), _ptr(pools.inst_pos), pools.m_cap, _ptr(new_fts), _ptr(new_pos),
                                          _ptr(pair_slot), _ptr(pair_inst), _ptr(pair_new), R, _ptr(out), self._stream()))
        return out

    def agent_frame_compact(self, pool_pos, pool_fts, slot, ids, n_ids, pose, radius):
        n, max_ids = ids.shape
        dev = ids.device
        rel = torch.empty((n, max_ids, 3), dtype=torch.float32, device=dev)
        fts = torch.empty((n, max_ids, FTS), dtype=torch.float32, device=dev)
        kept = torch.empty((n, max_ids), dtype=torch.int32, device=dev)
        count = torch.zeros((n,), dtype=torch.int32, device=dev)
        self._ck(self.lib.d3d_agent_frame_compact(_ptr(pool_pos), _ptr(pool_fts), pool_pos.shape[1], _ptr(slot), _ptr(ids), _ptr(n_ids),
                                                  n, max_ids, _ptr(pose), radius, _ptr(rel), _ptr(fts), _ptr(kept), _ptr(count),
                                                  self._stream()))
        return rel, fts, kept, count
