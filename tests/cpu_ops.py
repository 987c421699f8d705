"""TEST-ONLY numpy emulation of `dynam3d_amd.ops.HipOps`, built on the oracle (oracle/geometry.py).

It lets `pytest -m "not gpu"` exercise the REAL host orchestration (dynam3d_amd/feature_fields.py) and
the REAL C++ bookkeeping (csrc/ff_state.cpp, built CPU-only) in the GPU-less build container; each
method mirrors one C-ABI kernel's contract.  It is never importable from the product package and the
product never selects it: `Feature_Fields(ops=None)` always instantiates HipOps and raises without
libdynam3d_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from dynam3d_amd.build import build_host_state
from dynam3d_amd.ff_plan import FFDevOps
from dynam3d_amd.ops import FTS, CameraTables, Pools
from oracle import geometry as G

F32 = np.float32


class _Pose:
    def __init__(self, p):
        self.cam = p[0:3]
        self.cos_h, self.sin_h, self.cos_nh, self.sin_nh, self.heading = p[3], p[4], p[5], p[6], p[7]


def _np(t):
    return t.numpy() if isinstance(t, torch.Tensor) else t


class CpuOps(FFDevOps):
    """(FFDevOps: the device planner's d3d_ffdev_* entry points -- here the SAME source, csrc/ff_plan.h, compiled over host arrays into
    the CPU-only test library.)"""
    name = "cpu-emulation"

    def __init__(self):
        self.lib = C.CDLL(build_host_state())
        self.lib.d3d_last_error.restype = C.c_char_p

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"libd3d_ffstate_host: {self.lib.d3d_last_error().decode()} (code {rc})")

    # a1 / a2
    def preprocess_depth(self, depth, lo=0.0, hi=10.0):
        d = _np(depth)
        sh = d.shape
        return torch.from_numpy(G.preprocess_depth(d.reshape(sh[0], sh[1], sh[2], 1), (lo, hi)).reshape(sh))

    def resize_nearest_preprocess(self, depth, h, w, lo=0.0, hi=10.0):
        d = _np(depth)
        d = d.reshape(d.shape[0], d.shape[1], d.shape[2], 1)
        r = G.preprocess_depth(G.downsample_depth_nearest(d, (h, w)), (lo, hi))
        return torch.from_numpy(r[..., 0].copy())

    # a5
    def unproject_append(self, depth24, pose, slot, row_base, cam: CameraTables, pools: Pools):
        d, ps, sl, rb = _np(depth24), _np(pose), _np(slot), _np(row_base)
        txy, tz, d0 = _np(cam.tan_xy), _np(cam.tan_z), _np(cam.dir0)
        for e in range(d.shape[0]):
            p = _Pose(ps[e])
            dd = d[e].astype(F32)
            dx, dz = (dd * txy).astype(F32), (dd * tz).astype(F32)
            sc = (((dd * F32(cam.th)).astype(F32) * F32(2.0)).astype(F32) / F32(cam.W)).astype(F32)
            direction = G._pymod((d0 + p.heading).astype(F32), G.TWO_PI_F32)
            rx = ((dx * p.cos_h).astype(F32) - (dd * p.sin_h).astype(F32)).astype(F32)
            ry = ((dx * p.sin_h).astype(F32) + (dd * p.cos_h).astype(F32)).astype(F32)
            pos = np.stack([(rx + p.cam[0]).astype(F32), (ry + p.cam[1]).astype(F32), (dz + p.cam[2]).astype(F32)], -1)
            r0, P = int(rb[e]), d.shape[1]
            pools.rows_pos[sl[e], r0:r0 + P] = torch.from_numpy(pos)
            pools.rows_dir[sl[e], r0:r0 + P] = torch.from_numpy(direction)
            pools.rows_scale[sl[e], r0:r0 + P] = torch.from_numpy(sc)

    def append_fts(self, grid, slot, row_base, pools: Pools):
        sl, rb = _np(slot), _np(row_base)
        P = grid.shape[1]
        for e in range(grid.shape[0]):
            pools.rows_fts[sl[e], int(rb[e]):int(rb[e]) + P] = grid[e].to(torch.float16)

    def patch_3d_info(self, depth24, cam: CameraTables):
        outs = G.patch_3d_info(_np(depth24), cam.H, cam.W, cam.hfov, cam.vfov)
        return [torch.from_numpy(np.ascontiguousarray(o[..., 0])) for o in outs]

    # a4
    @staticmethod
    def _mask(points, depth_img, p: _Pose, intr, near, far, slack):
        fx, fy, cx, cy = (F32(v) for v in intr)
        Hd, Wd = depth_img.shape
        pts = points.astype(F32)
        px, py, pz = (pts[:, 0] - p.cam[0]).astype(F32), (pts[:, 1] - p.cam[1]).astype(F32), (pts[:, 2] - p.cam[2]).astype(F32)
        rx = ((px * p.cos_nh).astype(F32) - (py * p.sin_nh).astype(F32)).astype(F32)
        ry = ((px * p.sin_nh).astype(F32) + (py * p.cos_nh).astype(F32)).astype(F32)
        X, Y, Z = rx, (-pz).astype(F32), ry
        with np.errstate(all="ignore"):
            uf = (((fx * X).astype(F32) + (cx * Z).astype(F32)).astype(F32) / Z).astype(F32)
            vf = (((fy * Y).astype(F32) + (cy * Z).astype(F32)).astype(F32) / Z).astype(F32)
            ok = (uf > -1) & (uf < F32(Wd)) & (vf > -1) & (vf < F32(Hd)) & (Z >= F32(near)) & (Z <= F32(far))
        u = np.where(ok, np.trunc(np.where(ok, uf, 0)), 0).astype(np.int64)
        v = np.where(ok, np.trunc(np.where(ok, vf, 0)), 0).astype(np.int64)
        return ok & (Z < (depth_img.astype(F32)[v, u] + F32(slack)).astype(F32))

    def frustum_cull(self, pools: Pools, slot, n_rows, max_rows, depth, pose, intr, near, far, slack, hits, n_hits, mask=None):
        sl, nr, ps, dp = _np(slot), _np(n_rows), _np(pose), _np(depth)
        for e in range(len(sl)):
            pts = pools.rows_pos[sl[e], :nr[e]].numpy()
            m = self._mask(pts, dp[e], _Pose(ps[e]), intr, near, far, slack)
            idx = np.nonzero(m)[0]
            rng = np.random.default_rng(int(nr[e]))
            idx = rng.permutation(idx)                      # the kernel appends hits unordered
            hits[e, :len(idx)] = torch.from_numpy(idx.astype(np.int32))
            n_hits[e] = len(idx)
            mt = torch.from_numpy(m)
            pools.rows_pos[sl[e], :nr[e]][mt] = -10000.0
            pools.rows_fts[sl[e], :nr[e]][mt] = 0
            pools.rows_dir[sl[e], :nr[e]][mt] = 0
            pools.rows_scale[sl[e], :nr[e]][mt] = 0

    def frustum_mask(self, points, depth, pose_host, intr, near, far, slack):
        return torch.from_numpy(self._mask(_np(points), _np(depth), _Pose(np.asarray(pose_host, F32)), intr, near, far, slack).astype(np.uint8))

    # a6 after the segmenter
    def patch_segm_from_masks(self, masks, mask_off, h, w):
        mk = _np(masks)
        out = [G.patch_segm_from_masks(mk[mask_off[i]:mask_off[i + 1]], (h, w)) for i in range(len(mask_off) - 1)]
        segm = torch.from_numpy(np.stack(out))
        return segm, torch.tensor([int(o.max()) + 1 for o in out], dtype=torch.int32)

    # intrinsics / extrinsics path: same packed camera rows as the HIP entry points
    def frustum_cull_pinhole(self, pools: Pools, slot, n_rows, max_rows, depth, views, near, far, slack, hits, n_hits, mask=None):
        sl, nr, vw, dp = _np(slot), _np(n_rows), _np(views), _np(depth)
        for e in range(len(sl)):
            pts = pools.rows_pos[sl[e], :nr[e]].numpy()
            V = np.concatenate([vw[e, :12].reshape(3, 4), [[0, 0, 0, 1]]]).astype(F32)
            m = G.frustum_mask_pinhole(pts, dp[e], vw[e, 12:].reshape(3, 3), V, near, far, slack)
            idx = np.random.default_rng(int(nr[e])).permutation(np.nonzero(m)[0])
            hits[e, :len(idx)] = torch.from_numpy(idx.astype(np.int32))
            n_hits[e] = len(idx)
            mt = torch.from_numpy(m)
            pools.rows_pos[sl[e], :nr[e]][mt] = -10000.0
            pools.rows_fts[sl[e], :nr[e]][mt] = 0
            pools.rows_dir[sl[e], :nr[e]][mt] = 0
            pools.rows_scale[sl[e], :nr[e]][mt] = 0

    def frustum_mask_pinhole(self, points, depth, view_host, near, far, slack):
        vw = np.asarray(view_host, F32).reshape(-1)
        V = np.concatenate([vw[:12].reshape(3, 4), [[0, 0, 0, 1]]]).astype(F32)
        return torch.from_numpy(G.frustum_mask_pinhole(_np(points), _np(depth), vw[12:].reshape(3, 3), V, near, far, slack).astype(np.uint8))

    def unproject_pinhole_append(self, depth, cams, slot, row_base, h, w, input_width, pools: Pools):
        d, cm, sl, rb = _np(depth), _np(cams), _np(slot), _np(row_base)
        for e in range(d.shape[0]):
            K = np.array([[cm[e, 0], 0, cm[e, 2]], [0, cm[e, 1], cm[e, 3]], [0, 0, 1]])
            scale_tan, depth_scale, depth_trunc, _ = np.ascontiguousarray(cm[e, 16:18]).view(F32)
            pos, direction, sc = G.unproject_pinhole(d[e], K, cm[e, 4:13].reshape(3, 3), cm[e, 13:16], scale_tan, input_width,
                                                     depth_scale, depth_trunc, (h, w))
            r0, P = int(rb[e]), h * w
            pools.rows_pos[sl[e], r0:r0 + P] = torch.from_numpy(pos)
            pools.rows_dir[sl[e], r0:r0 + P] = torch.from_numpy(direction)
            pools.rows_scale[sl[e], r0:r0 + P] = torch.from_numpy(sc)

    # a8
    def knn(self, points, point_stride, n_points, queries, query_stride, n_queries, k, n_batch, max_queries, k_max):
        pts, q = _np(points).reshape(-1), _np(queries).reshape(-1)
        npts, nq, kk = _np(n_points), _np(n_queries), _np(k)
        d2 = np.full((n_batch, max_queries, k_max), np.inf, F32)
        idx = np.full((n_batch, max_queries, k_max), -1, np.int32)
        for b in range(n_batch):
            P_ = pts[b * point_stride: b * point_stride + npts[b] * 3].reshape(-1, 3)
            Q_ = q[b * query_stride: b * query_stride + nq[b] * 3].reshape(-1, 3)
            if kk[b] > 0 and nq[b] > 0:
                a, i = G.knn_bruteforce(P_, Q_, int(kk[b]))
                d2[b, :nq[b], :kk[b]], idx[b, :nq[b], :kk[b]] = a, i
        return torch.from_numpy(d2), torch.from_numpy(idx)

    # a7 / a10
    def group_stats7(self, pools: Pools, tok_slot, tok_row, grp_off, G_, cell_len, inst_pos=None, grp_slot=None, grp_inst=None):
        ts, tr, off = _np(tok_slot), _np(tok_row), _np(grp_off)
        T = len(tr)
        cen = np.full((G_, 3), np.nan, F32)
        cell = np.zeros((G_, 3), np.int32)
        geom = np.zeros((T, 7), F32)
        rp, rd, rs = pools.rows_pos.numpy(), pools.rows_dir.numpy(), pools.rows_scale.numpy()
        for g in range(G_):
            a, b = off[g], off[g + 1]
            if b == a:
                continue
            pos = rp[ts[a:b], tr[a:b]]
            cen[g] = G.mean_rows_f64(pos)
            with np.errstate(invalid="ignore"):
                cell[g] = np.floor(cen[g] / np.array(cell_len, F32)).astype(np.int32)
            geom[a:b] = G.segment_geometry(pos, rd[ts[a:b], tr[a:b]], rs[ts[a:b], tr[a:b]], cen[g])
            if inst_pos is not None and _np(grp_inst)[g] >= 0:
                inst_pos[int(_np(grp_slot)[g]), int(_np(grp_inst)[g])] = torch.from_numpy(cen[g])
        return torch.from_numpy(cen), torch.from_numpy(cell), torch.from_numpy(geom)

    # a11
    def group_stats4(self, pools: Pools, tok_slot, tok_inst, grp_off, grp_mode, grp_slot, grp_zone_row, G_, cell_len):
        ts, ti, off, mode, gs, gr = (_np(x) for x in (tok_slot, tok_inst, grp_off, grp_mode, grp_slot, grp_zone_row))
        geom = np.zeros((len(ti), 4), F32)
        ip = pools.inst_pos.numpy()
        for g in range(G_):
            a, b = off[g], off[g + 1]
            pos = ip[ts[a:b], ti[a:b]].reshape(-1, 3)
            if mode[g] == 1:
                pos = G.zone_cell_centre(pos, cell_len)
            cen = G.mean_rows_f64(pos) if b > a else np.full(3, np.nan, F32)
            pools.zone_pos[int(gs[g]), int(gr[g])] = torch.from_numpy(cen)
            if b > a:
                p = pos.astype(F32)
                n2 = (((p[:, 0] * p[:, 0]).astype(F32) + (p[:, 1] * p[:, 1]).astype(F32)).astype(F32) + (p[:, 2] * p[:, 2]).astype(F32)).astype(F32)
                geom[a:b, :3] = (p - cen[None]).astype(F32)
                geom[a:b, 3] = np.sqrt(n2)
        return torch.from_numpy(geom)

    # row movers
    def gather_fts(self, pools: Pools, tok_slot, tok_row):
        return pools.rows_fts[tok_slot.long(), tok_row.long()].float()

    def gather_rows(self, pool, slot, row):
        return pool[slot.long(), row.long()].clone()

    def scatter_rows(self, pool, slot, row, src, src_row=None):
        s = src if src_row is None else src[src_row.long()]
        ok = row >= 0                                       # row -1 = nothing to store (device-planned scatters)
        pool[slot.long()[ok], row.long()[ok]] = s[ok]

    def fill_rows(self, pool, slot, row, value):
        pool[slot.long(), row.long()] = value

    # a9
    def merge_input(self, pools: Pools, new_fts, new_pos, pair_slot, pair_inst, pair_new):
        s, i, n = pair_slot.long(), pair_inst.long(), pair_new.long()
        return torch.cat([pools.inst_fts[s, i], new_fts[n], new_pos[n] - pools.inst_pos[s, i]], dim=-1).contiguous()

    # a12
    def agent_frame_compact(self, pool_pos, pool_fts, slot, ids, n_ids, pose, radius):
        sl, idn, nn, ps = _np(slot), _np(ids), _np(n_ids), _np(pose)
        n, mx = idn.shape
        rel = torch.zeros((n, mx, 3))
        fts = torch.zeros((n, mx, FTS))
        kept = torch.zeros((n, mx), dtype=torch.int32)
        count = torch.zeros((n,), dtype=torch.int32)
        for e in range(n):
            id_e = idn[e, :nn[e]]
            p = _Pose(ps[e])
            pts = pool_pos[sl[e]].numpy()[id_e].reshape(-1, 3)
            px, py, pz = (pts[:, 0] - p.cam[0]).astype(F32), (pts[:, 1] - p.cam[1]).astype(F32), (pts[:, 2] - p.cam[2]).astype(F32)
            rx = ((px * p.cos_nh).astype(F32) - (py * p.sin_nh).astype(F32)).astype(F32)
            ry = ((px * p.sin_nh).astype(F32) + (py * p.cos_nh).astype(F32)).astype(F32)
            with np.errstate(over="ignore", invalid="ignore"):
                n2 = (((rx * rx).astype(F32) + (ry * ry).astype(F32)).astype(F32) + (pz * pz).astype(F32)).astype(F32)
                keep = np.sqrt(n2) <= F32(radius)
            c = int(keep.sum())
            rel[e, :c] = torch.from_numpy(np.stack([rx, ry, pz], -1)[keep])
            fts[e, :c] = pool_fts[sl[e]][torch.from_numpy(id_e[keep].astype(np.int64))]
            kept[e, :c] = torch.from_numpy(id_e[keep].astype(np.int32))
            count[e] = c
        return rel, fts, kept, count
