"""Novel-view feature rendering of the Pretrain `Feature_Fields`:
`render_view_3d_patch` PRE-FF:494-625 with `get_rays_habitat` (408-422), `patch_to_nerf_encode` (477-491) and
`raw2feature` (446-474), batched over environments on the device-resident patch pools.

    rays (d3d_rays_habitat) -> k=4 KNN of 72 144 samples vs the stored patches (d3d_knn) -> importance top-8 per ray
    (d3d_ray_topk) -> neighbour gather + 6-d geometry + Linear(6,768)+LN + fp16 add (d3d_render_embed) ->
    Linear(3072,768)+LN -> tcnn encoder (768-768-768-769, LeakyReLU) -> +residual -> tcnn decoder -> alpha compositing
    (d3d_composite).  All GEMMs are fp16 MFMA (d3d_gemm_nt)."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import f32, i32, i64, vp
from .hip_dense import HipDense
from .ops import FTS, Pools
from .tcnn import Network

_lib.register("d3d_rays_habitat", [vp, vp, vp, vp, i32, i32, i32, vp, vp])
_lib.register("d3d_rays_pinhole", [vp, vp, i32, i32, i32, i32, vp, vp])
_lib.register("d3d_ray_topk", [vp, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp])
_lib.register("d3d_render_embed", [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, f32, vp, vp, vp, vp])
_lib.register("d3d_composite", [vp, i64, vp, i64, vp, vp, i32, i32, i32, vp, vp, vp])


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FieldRenderer:
    RADIUS_KNN = True

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", view_hw=(12, 12), n_samples=501, n_importance=8, k=4, radius=1.0,
                 near=0.0, far=10.0, hfov=90.0, vfov=90.0, width=768):
        self.lib = _lib.load()
        self.hd = HipDense()
        self.dev = torch.device(device)
        self.H, self.W = view_hw
        self.N, self.n_imp, self.k, self.radius, self.near, self.far = n_samples, n_importance, k, radius, near, far
        f = lambda n: sd[n].detach().to(self.dev, torch.float32).contiguous()
        h = lambda n: sd[n].detach().to(self.dev, torch.float16).contiguous()
        self.w6, self.b6 = f("patch_to_nerf_position_embedding.0.weight"), f("patch_to_nerf_position_embedding.0.bias")
        self.ln6 = (f("patch_to_nerf_position_embedding.1.weight"), f("patch_to_nerf_position_embedding.1.bias"))
        self.agg_w, self.agg_b = h("aggregate_patch_to_nerf_encoder.0.weight"), h("aggregate_patch_to_nerf_encoder.0.bias")
        self.agg_ln = (f("aggregate_patch_to_nerf_encoder.1.weight"), f("aggregate_patch_to_nerf_encoder.1.bias"))
        cfg = lambda out_act, nh: {"otype": "CutlassMLP", "activation": "LeakyReLU", "output_activation": out_act, "n_neurons": width, "n_hidden_layers": nh}
        def net(name, n_out, out_act):
            # a tinycudann checkpoint holds ONE flat `<name>.params` (PRE-FF:221-243); the harness / synthetic weights use per-layer matrices
            if name + ".params" in sd:
                return Network.from_flat_params(width, n_out, cfg(out_act, 2), sd[name + ".params"], device=device)
            return Network(width, n_out, cfg(out_act, 2), [sd[f"{name}.layers.{i}.weight"] for i in range(3)], device)
        self.encoder, self.decoder = net("nerf_encoder", width + 1, "LeakyReLU"), net("nerf_decoder", width, "None")
        for m in (self.encoder, self.decoder):
            m.requires_grad_(False)                                 # the renderer is the inference path: one C call per network
        # ray tables (PRE-FF:408-422): float64 linspace, float32 tangents
        R = self.H * self.W
        hW, hH = self.W // 2, self.H // 2
        tan_xy = np.array(([[i / hW + 1 / self.W] for i in range(-hW, hW)]) * self.H, np.float32) * math.tan(np.deg2rad(hfov) / 2.0)
        tan_z = np.array([[i / hH - 1 / self.H for i in range(hH, -hH, -1)]] * self.W, np.float32).T.reshape((-1, 1)) * math.tan(np.deg2rad(vfov) / 2.0)
        rel_y = np.linspace(near, far, n_samples)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(self.dev)
        self.tan_xy, self.tan_z = t(tan_xy.reshape(-1), np.float32), t(tan_z.reshape(-1), np.float32)
        self.rel_y64 = t(rel_y, np.float64)
        self.rel_dir = t((-np.arctan(tan_xy)).reshape(-1), np.float32)
        self.rel_dist16 = t(rel_y.astype(np.float16).astype(np.float32), np.float32)       # PRE-FF:620 stores rel_dist as fp16
        self.R = R

    @staticmethod
    def _stream():
        return _lib.current_stream_ptr()

    @torch.no_grad()
    def _pinhole_tables(self, fx: float, fy: float):
        """get_rays (PRE-FF:390-405) for the view-sized intrinsics: depth samples near + spacing*(i+1) stored as float32 images,
        unprojected by Open3D in double; rel_direction = -arctan(x/z) of the last sample; rel_dist = z (later cast to fp16)."""
        key = (float(fx), float(fy))
        if getattr(self, "_pin_key", None) != key:
            spacing = (self.far - self.near) / self.N
            z = np.array([np.float32(self.near + spacing * (i + 1)) for i in range(self.N)], np.float64)
            cx = self.W / 2
            x_last = (np.tile(np.arange(self.W, dtype=np.float64), self.H) - cx) * z[-1] / fx
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(self.dev)
            self._pin = dict(z=t(z, np.float64), rel_dir=t(-np.arctan(x_last / z[-1]), np.float32),
                             rel_dist16=t(z.astype(np.float16).astype(np.float32), np.float32))
            self._pin_key = key
        return self._pin

    def front(self, pools: Pools, slots: Sequence[int], n_rows: Sequence[int], batch_position, batch_heading, ops, want_geom6=False,
              batch_rot=None, batch_trans=None, view_intrinsic=None, raw_features=False):
        """Everything in front of the networks (PRE-FF:494-615): rays, k = 4 KNN of every depth sample against the stored patches, importance
        top-8 per ray, neighbour gather + 6-d relative geometry [+ position embedding].  Returns a dict: s16 (n_rays * S, K * 768) fp16 =
        neighbour feature + Linear(6, 768) + LN of its geometry (fp16 add) -- or, with `raw_features`, the gathered features alone (the
        training step evaluates the position embedding differentiably itself, train_render.py) --, geom6 (n_rays * S * K, 6) when asked for,
        topk (n_rays, S), sidx, n_ranked, sample_xyz, ray, rel_dist16."""
        B, R, N, S, K = len(slots), self.R, self.N, self.n_imp, self.k
        dev, lib, st = self.dev, self.lib, self._stream
        i32t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)
        ray = torch.empty((B, R * N, 3), dtype=torch.float32, device=dev)
        if batch_rot is not None:
            fx, fy = view_intrinsic
            tab = self._pinhole_tables(fx, fy)
            cam16 = np.zeros((B, 16), np.float64)
            heads = []
            for b in range(B):
                Rm, T = np.asarray(batch_rot[b], np.float64).reshape(3, 3), np.asarray(batch_trans[b], np.float64).reshape(3)
                cam16[b] = [fx, fy, self.W / 2, self.H / 2, *Rm.reshape(-1), *T]
                fwd = Rm @ np.array([0.0, 0.0, 1.0]) + T                   # PRE-FF:512-515: heading of the WORLD point R@[0,0,1]+T
                xy = max(math.sqrt(fwd[0] ** 2 + fwd[1] ** 2), 1e-4)
                h = -math.asin(fwd[0] / xy) - (math.pi if fwd[1] < 0 else 0.0)
                heads.append(float(np.float32(h)))                         # torch.tensor(..., dtype=float32)
            _lib.check(lib.d3d_rays_pinhole(_p(tab["z"]), _p(torch.from_numpy(cam16).to(dev)), B, self.H, self.W, N, _p(ray), st()))
            rel_dir, rel_dist16, batch_heading = tab["rel_dir"], tab["rel_dist16"], heads
        else:
            pose64 = np.array([[p[0], -p[2], p[1], math.cos(h), math.sin(h)] for p, h in zip(batch_position, batch_heading)], np.float64)
            _lib.check(lib.d3d_rays_habitat(_p(self.rel_y64), _p(self.tan_xy), _p(self.tan_z), _p(torch.from_numpy(pose64).to(dev)), B, R, N, _p(ray), st()))
            rel_dir, rel_dist16 = self.rel_dir, self.rel_dist16
        pose3 = np.array([[np.float32(math.cos(-h)), np.float32(math.sin(-h)), np.float32(h)] for h in batch_heading], np.float32)
        ident = all(s == i for i, s in enumerate(slots))
        pts = pools.rows_pos if ident else pools.rows_pos.index_select(0, i32t(slots).long()).contiguous()
        # PRE-FF:540: neighbours at >= radius are discarded by d3d_ray_topk right behind this query, so the radius-limited kernel serves
        # (identical inside the radius; RADIUS_KNN = False: the exact brute-force d3d_knn, the A/B baseline)
        kw = {"radius": self.radius} if self.RADIUS_KNN else {}
        d2, idx = ops.knn(pts, pools.n_cap * 3, i32t(n_rows), ray, R * N * 3, i32t([R * N] * B), i32t([K] * B), B, R * N, K, **kw)
        n_rays = B * R
        topk = torch.empty((n_rays, S), dtype=torch.int32, device=dev)
        sidx = torch.empty((n_rays, S, K), dtype=torch.int32, device=dev)
        n_ranked = torch.empty((n_rays,), dtype=torch.int32, device=dev)
        _lib.check(lib.d3d_ray_topk(_p(d2), _p(idx), n_rays, N, K, self.radius, S, _p(topk), _p(sidx), _p(n_ranked), st()))
        ray_env = i32t(np.repeat(np.arange(B), R))
        ray_slot = i32t(np.repeat(np.asarray(slots), R))
        s16 = torch.empty((n_rays * S, K * FTS), dtype=torch.float16, device=dev)
        sample_xyz = torch.empty((n_rays, S, 3), dtype=torch.float32, device=dev)
        geom6 = torch.empty((n_rays * S * K, 6), dtype=torch.float32, device=dev) if want_geom6 else None
        if raw_features:
            # zero Linear + zero LayerNorm bias: LN(0) * gain + 0 = 0, and the kernel's fp16 add leaves the gathered features untouched
            if getattr(self, "_zero6", None) is None:
                self._zero6 = (torch.zeros_like(self.w6), torch.zeros_like(self.b6), torch.ones_like(self.ln6[0]), torch.zeros_like(self.ln6[1]))
            w6, b6, g6, be6 = self._zero6
        else:
            w6, b6, g6, be6 = self.w6, self.b6, self.ln6[0], self.ln6[1]
        _lib.check(lib.d3d_render_embed(_p(pools.rows_pos), _p(pools.rows_dir), _p(pools.rows_scale), _p(pools.rows_fts), pools.n_cap, _p(ray_slot),
                                        _p(ray_env), _p(ray), _p(topk), _p(sidx), _p(torch.from_numpy(pose3).to(dev)), _p(rel_dir), n_rays, R, N, S, K,
                                        self.far, _p(w6), _p(b6), _p(g6), _p(be6), 1e-12, _p(s16), _p(geom6), _p(sample_xyz), st()))
        return dict(s16=s16, geom6=geom6, topk=topk, sidx=sidx, n_ranked=n_ranked, sample_xyz=sample_xyz, ray=ray, rel_dist16=rel_dist16, n_rays=n_rays, B=B)

    def render(self, pools: Pools, slots: Sequence[int], n_rows: Sequence[int], batch_position, batch_heading, ops, debug=False,
               batch_rot=None, batch_trans=None, view_intrinsic=None):
        """-> (features (B,H,W,768) f32 unit-norm, positions (B,H,W,3), depth (B,H,W)[, debug dict]).
        Habitat mode: `batch_position` / `batch_heading`.  Intrinsics mode (PRE-FF:505-515, 532-536): camera->world `batch_rot` /
        `batch_trans` per env and the view-sized (fx, fy) of the last update."""
        fr = self.front(pools, slots, n_rows, batch_position, batch_heading, ops, want_geom6=debug, batch_rot=batch_rot, batch_trans=batch_trans,
                        view_intrinsic=view_intrinsic)
        B, R, N, S, K = fr["B"], self.R, self.N, self.n_imp, self.k
        dev, lib, st = self.dev, self.lib, self._stream
        s16, geom6, topk, sidx, n_ranked, sample_xyz, ray, rel_dist16, n_rays = (fr[k] for k in ("s16", "geom6", "topk", "sidx", "n_ranked", "sample_xyz",
                                                                                             "ray", "rel_dist16", "n_rays"))
        x = self.hd.gemm(s16, self.agg_w, self.agg_b, None, "bias")                              # PRE-FF:483 Linear(3072,768)
        x = self.hd.layer_norm(x, self.agg_ln[0], self.agg_ln[1], 1e-12)
        enc = self.encoder(x)                                                                    # (M,769) fp16   PRE-FF:484
        dens = enc[:, FTS]
        y = (enc[:, :FTS].float() + x.float()).to(torch.float16)                                 # residual PRE-FF:487
        out = self.decoder(y).contiguous()                                                       # (M,768) fp16   PRE-FF:488
        fmap = torch.empty((n_rays, FTS), dtype=torch.float32, device=dev)
        depth = torch.empty((n_rays,), dtype=torch.float32, device=dev)
        _lib.check(lib.d3d_composite(_p(out), out.stride(0), _p(dens), dens.stride(0), _p(rel_dist16), _p(topk), n_rays, N, S, _p(fmap), _p(depth), st()))
        res = (fmap.view(B, self.H, self.W, FTS), sample_xyz[:, 0].reshape(B, self.H, self.W, 3), depth.view(B, self.H, self.W))
        if debug:
            return res + (dict(topk=topk.view(B, R, S), sidx=sidx.view(B, R, S, K), n_ranked=n_ranked.view(B, R), geom6=geom6.view(B, R, S, K, 6),
                               density=dens.float().view(B, R, S), feat=out.float().view(B, R, S, FTS), ray=ray.view(B, R, N, 3)),)
        return res
