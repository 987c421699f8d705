"""CPU restatement of the Pretrain variant's novel-view feature rendering:
`get_rays_habitat` PRE-FF:408-422, `render_view_3d_patch` PRE-FF:494-625 (habitat mode, no GT labels),
`patch_to_nerf_encode` PRE-FF:477-491, `raw2feature` PRE-FF:446-474.   TEST INFRASTRUCTURE.

Third-party arithmetic that is absent from /root/reference and therefore DEFINED here ("parity unpinned"):
  * torch_kdtree  -> ascending (d^2, index) brute force (oracle/geometry.py::knn_bruteforce)
  * tinycudann CutlassMLP -> bias-free layers, y = act(x W^T), fp16 weights and activations, fp32 accumulation,
    result rounded to fp16 after every layer, LeakyReLU slope 0.01 (oracle/ref_harness.py::TcnnStub does the same)
  * torch.topk tie order (PRE-FF:555) -> descending density, ties -> lowest sample index.  Rays with fewer than
    N_importance strictly-ranked samples depend on that unpinned order in the reference and are excluded when
    comparing against reference-generated goldens.
Reference quirk reproduced (R1): the in-place rotation at PRE-FF:594-597 reads channel 0 AFTER overwriting it
(`sample_ft_neighbor_x` is a view), so y' = x'*sin(-c) + y*cos(-c).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import geometry as G

F32 = np.float32


def rays_habitat(H=12, W=12, near=0.0, far=10.0, n_samples=501, hfov=90.0, vfov=90.0):
    """PRE-FF:408-422.  Returns rel_x, rel_y, rel_z (R,N) float64, rel_direction (R,1) f32."""
    rel_y = np.expand_dims(np.linspace(near, far, n_samples), axis=0).repeat(H * W, axis=0)
    hW, hH = W // 2, H // 2
    tan_xy = np.array(([[i / hW + 1 / W] for i in range(-hW, hW)]) * H, F32) * math.tan(np.deg2rad(hfov) / 2.0)
    rel_direction = -np.arctan(tan_xy)
    rel_x = rel_y * tan_xy
    tz = np.array([[i / hH - 1 / H for i in range(hH, -hH, -1)]] * W, F32).T.reshape((-1, 1)) * math.tan(np.deg2rad(vfov) / 2.0)
    rel_z = rel_y * tz
    return rel_x, rel_y, rel_z, rel_direction.astype(F32)


def world_rays(rel_x, rel_y, rel_z, position_habitat, heading) -> np.ndarray:
    """PRE-FF:524-530: float64 rotation + translation, rounded once to float32.  -> (R,N,3)."""
    cx, cy, cz = position_habitat[0], -position_habitat[2], position_habitat[1]
    c, s = math.cos(heading), math.sin(heading)
    x = rel_x * c - rel_y * s + cx
    y = rel_x * s + rel_y * c + cy
    z = rel_z + cz
    return np.stack([x, y, z], -1).astype(F32)


def ray_topk(d2: np.ndarray, idx: np.ndarray, R: int, N: int, radius=1.0, n_imp=8):
    """PRE-FF:543-556.  d2/idx (R*N,k).  Returns masked idx (R,N,k), topk_inds (R,n_imp), n_ranked (R,) = number of
    samples with at least one neighbour inside the radius (rays with n_ranked < n_imp hit the unpinned tie order)."""
    d = np.sqrt(d2.astype(F32)).astype(F32)
    far_ = d >= F32(radius)
    idx = np.where(far_, -1, idx).reshape(R, N, -1)
    d = np.where(far_, F32(radius), d).reshape(R, N, -1)
    tmp = d[..., 0]
    for j in range(1, d.shape[-1]):
        tmp = (tmp + d[..., j]).astype(F32)
    dens = (F32(1.0) / tmp).astype(F32)
    order = np.argsort(-dens, axis=1, kind="stable")[:, :n_imp]
    n_ranked = (idx.max(-1) >= 0).sum(1)
    return idx, order.astype(np.int64), n_ranked


def neighbour_geometry(patch_pos, patch_dir, patch_scale, sample_xyz, idx, heading, rel_direction, far=10.0):
    """PRE-FF:586-611.  sample_xyz (R,S,3), idx (R,S,k) with -1 = none.  -> (R,S,k,6) f32."""
    R, S, k = idx.shape
    safe = np.where(idx < 0, 0, idx)
    out = np.zeros((R, S, k, 6), F32)
    dlt = (patch_pos[safe] - sample_xyz[:, :, None, :]).astype(F32)
    c, s = F32(math.cos(-heading)), F32(math.sin(-heading))
    x, y = dlt[..., 0], dlt[..., 1]
    xr = ((x * c).astype(F32) - (y * s).astype(F32)).astype(F32)
    yr = ((xr * s).astype(F32) + (y * c).astype(F32)).astype(F32)          # quirk R1: uses the ROTATED x
    out[..., 0], out[..., 1], out[..., 2] = xr, yr, dlt[..., 2]
    none = idx < 0
    out[..., :3][none] = F32(far)
    pd = (patch_dir.astype(F32) - F32(heading)).astype(F32)                 # PRE-FF:519
    ang = (pd[safe] - rel_direction.reshape(R, 1, 1).astype(F32)).astype(F32)
    out[..., 3], out[..., 4] = np.sin(ang), np.cos(ang)
    out[..., 5] = patch_scale.astype(F32)[safe]
    out[..., 3:][none] = 0
    return out


def tcnn_mlp(x: torch.Tensor, weights, act="LeakyReLU", out_act="None") -> torch.Tensor:
    """CutlassMLP stand-in: fp16 weights/activations, fp32 accumulation, fp16 store per layer."""
    h = x.to(torch.float16)
    for i, w in enumerate(weights):
        y = h.float() @ w.to(torch.float16).float().t()
        a = act if i < len(weights) - 1 else out_act
        if a == "LeakyReLU":
            y = F.leaky_relu(y, 0.01)
        h = y.to(torch.float16)
    return h


def nerf_encode(emb16: torch.Tensor, geom6: torch.Tensor, sd: Dict[str, torch.Tensor], n_imp=8):
    """PRE-FF:477-491.  emb16 (R,S,k,768) f16, geom6 (R,S,k,6) f32 -> features (R,S,768) f16, density (R,S) f16."""
    W = emb16.shape[-1]
    k = emb16.shape[-2]
    e = emb16.reshape(-1, W * k).to(torch.float16)
    g = F.linear(geom6, sd["patch_to_nerf_position_embedding.0.weight"].float(), sd["patch_to_nerf_position_embedding.0.bias"].float())
    g = F.layer_norm(g, (W,), sd["patch_to_nerf_position_embedding.1.weight"].float(), sd["patch_to_nerf_position_embedding.1.bias"].float(), 1e-12)
    g = g.reshape(-1, W * k).to(torch.float16)
    s = (e + g).float()                                                      # fp16 add, then the fp32 Linear (F12)
    x = F.linear(s, sd["aggregate_patch_to_nerf_encoder.0.weight"].float(), sd["aggregate_patch_to_nerf_encoder.0.bias"].float())
    x = F.layer_norm(x, (W,), sd["aggregate_patch_to_nerf_encoder.1.weight"].float(), sd["aggregate_patch_to_nerf_encoder.1.bias"].float(), 1e-12)
    enc = tcnn_mlp(x, [sd[f"nerf_encoder.layers.{i}.weight"] for i in range(3)], "LeakyReLU", "LeakyReLU")
    feat, dens = enc[:, :-1], enc[:, -1]
    y = feat.float() + x                                                     # residual in fp32 (fp16 + fp32 promotes)
    out = tcnn_mlp(y, [sd[f"nerf_decoder.layers.{i}.weight"] for i in range(3)], "LeakyReLU", "None")
    return out.view(-1, n_imp, W), dens.reshape(-1, n_imp)


def raw2feature(feat16: torch.Tensor, dens16: torch.Tensor, rel_dist: np.ndarray, topk: np.ndarray):
    """PRE-FF:446-474 evaluated in float32 on the fp16 inputs.  feat16 (R,S,768), dens16 (R,S), rel_dist (R,N) (stored as
    fp16 by the reference, PRE-FF:620), topk (R,S).  -> feature_map (R,768) unit-norm, depth_map (R,)."""
    rd = torch.from_numpy(rel_dist.astype(np.float16).astype(np.float32))
    tk = torch.from_numpy(topk.astype(np.int64))
    sd_ = F.softplus(dens16.float())
    dists = torch.abs(rd[..., 1:] - rd[..., :-1])
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    density = torch.zeros_like(rd).scatter(1, tk, sd_)
    alpha = 1.0 - torch.exp(-F.relu(density) * dists)
    weights = alpha * torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1)), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    sw = torch.gather(weights, 1, tk)
    fm = torch.sum(sw[..., None] * feat16.float(), -2)
    fm = fm / torch.clamp(torch.linalg.norm(fm, dim=-1, keepdim=True), min=1e-7)
    depth = torch.sum(weights * rd, -1) / torch.clamp(torch.sum(weights, -1), min=1e-7)
    return fm, depth


@torch.no_grad()
def render_view(patch_pos, patch_dir, patch_scale, patch_fts16, sd, position_habitat, heading, H=12, W=12, n_samples=501,
                n_imp=8, k=4, radius=1.0, near=0.0, far=10.0, hfov=90.0, vfov=90.0):
    """One environment.  Returns dict(feature_map (H,W,768), positions (H,W,3), depth (H,W), topk, n_ranked, geom6, density)."""
    rel_x, rel_y, rel_z, rel_dir = rays_habitat(H, W, near, far, n_samples, hfov, vfov)
    R = H * W
    ray = world_rays(rel_x, rel_y, rel_z, position_habitat, heading)
    d2, idx = G.knn_bruteforce(patch_pos, ray.reshape(-1, 3), k)
    idx_m, topk, n_ranked = ray_topk(d2, idx, R, n_samples, radius, n_imp)
    sample_xyz = np.take_along_axis(ray, topk[..., None].repeat(3, -1), 1)               # (R,S,3)
    sidx = np.take_along_axis(idx_m, topk[..., None].repeat(k, -1), 1)                   # KNN #2 == rows of KNN #1
    geom6 = neighbour_geometry(patch_pos, patch_dir, patch_scale, sample_xyz, sidx, heading, rel_dir, far)
    emb = patch_fts16[np.where(sidx < 0, 0, sidx)].astype(np.float16)
    emb[sidx < 0] = 0
    feat, dens = nerf_encode(torch.from_numpy(emb), torch.from_numpy(geom6), sd, n_imp)
    fm, depth = raw2feature(feat, dens, rel_y, topk)
    return dict(feature_map=fm.numpy().reshape(H, W, -1), positions=sample_xyz[:, 0].reshape(H, W, 3), depth=depth.numpy().reshape(H, W),
                topk=topk, n_ranked=n_ranked, geom6=geom6, density=dens.float().numpy(), feat=feat.float().numpy(), sidx=sidx)


def render_view_pinhole(patch_pos, patch_dir, patch_scale, patch_fts16, sd, rot, trans, fx, fy, H=12, W=12, n_samples=501, n_imp=8, k=4,
                        radius=1.0, near=0.0, far=10.0):
    """Intrinsics mode of render_view_3d_patch (PRE-FF:505-515, 532-536), one environment: rays from get_rays(init_camera_intrinsic)
    (oracle/geometry.py::rays_pinhole; Open3D restated, PARITY UNPINNED), world = R @ rel + T in double then float32, camera heading
    = float32(get_heading_angle(R @ [0,0,1] + T)) -- of the world POINT, translation included, as the reference does -- and the
    rest as `render_view`.  The ray's own direction enters the neighbour geometry as float64 in the reference (PRE-FF:603-605) and
    as float32 here: a 6e-8 rad difference, inside the test tolerance."""
    rel, rel_dir, rel_dist = G.rays_pinhole(fx, fy, H, W, near, far, n_samples)
    R_ = H * W
    Rm, T = np.asarray(rot, np.float64).reshape(3, 3), np.asarray(trans, np.float64).reshape(3, 1)
    ray = (Rm @ rel.reshape(-1, 3).T + T).T.astype(F32).reshape(R_, n_samples, 3)
    fwd = (Rm @ np.array([[0.0, 0.0, 1.0]]).T + T).T
    heading = float(F32(G.heading_angle(fwd)[0]))
    d2, idx = G.knn_bruteforce(patch_pos, ray.reshape(-1, 3), k)
    idx_m, topk, n_ranked = ray_topk(d2, idx, R_, n_samples, radius, n_imp)
    sample_xyz = np.take_along_axis(ray, topk[..., None].repeat(3, -1), 1)
    sidx = np.take_along_axis(idx_m, topk[..., None].repeat(k, -1), 1)
    geom6 = neighbour_geometry(patch_pos, patch_dir, patch_scale, sample_xyz, sidx, heading, rel_dir.astype(F32), far)
    emb = patch_fts16[np.where(sidx < 0, 0, sidx)].astype(np.float16)
    emb[sidx < 0] = 0
    feat, dens = nerf_encode(torch.from_numpy(emb), torch.from_numpy(geom6), sd, n_imp)
    fm, depth = raw2feature(feat, dens, rel_dist, topk)
    return dict(feature_map=fm.numpy().reshape(H, W, -1), positions=sample_xyz[:, 0].reshape(H, W, 3), depth=depth.numpy().reshape(H, W),
                topk=topk, n_ranked=n_ranked, heading=heading, ray=ray)
