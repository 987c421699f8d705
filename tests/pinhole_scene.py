"""Synthetic posed RGB-D 'dataset' frames for the intrinsics / extrinsics path: a camera orbiting inside a
box-shaped room, raw uint16-style depth in millimetres, pinhole intrinsics, camera->world (R, T) and the world->camera matrix."""
import math

import numpy as np


def look_rotation(yaw: float, pitch: float) -> np.ndarray:
    """camera -> world rotation; camera looks along +z, x right, y down; world z is up."""
    cz, sz, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    fwd = np.array([cz * cp, sz * cp, sp])
    right = np.array([sz, -cz, 0.0])
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd], 1)


def room_depth(K, R, T, hw, half=(3.0, 2.5, 1.4)) -> np.ndarray:
    """Depth (metres along the optical axis) of an axis-aligned box room seen from T with rotation R."""
    H, W = hw
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    d_cam = np.stack([(jj - K[0, 2]) / K[0, 0], (ii - K[1, 2]) / K[1, 1], np.ones_like(jj, float)], -1)
    d_w = d_cam @ R.T
    t = np.full((H, W), np.inf)
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = (sgn * half[ax] - T[ax]) / d_w[..., ax]
            tt[tt <= 1e-6] = np.inf
            t = np.minimum(t, tt)
    return t


def frames(B=2, V=2, steps=3, hw=(60, 80), seed=0):
    """Yields per step: depth_raw (B,V,H,W) float32 millimetres (some exact zeros), depth_m (B,V,H,W) metres for the cull,
    intrinsics (B,V,3,3), rot (B,V,3,3), trans (B,V,3,1), extrinsic (B,V,4,4) world->camera, patch_segm (B*V,1,24,24)."""
    rng = np.random.default_rng(seed)
    H, W = hw
    K = np.array([[0.8 * W, 0, W / 2 - 0.5], [0, 0.8 * W, H / 2 - 0.5], [0, 0, 1.0]])
    yaw0 = rng.uniform(0, 2 * math.pi, B)
    for t in range(steps):
        out = dict(depth_raw=np.zeros((B, V, H, W), np.float32), depth_m=np.zeros((B, V, H, W), np.float32), intrinsics=np.zeros((B, V, 3, 3)),
                   rot=np.zeros((B, V, 3, 3)), trans=np.zeros((B, V, 3, 1)), extrinsic=np.zeros((B, V, 4, 4)))
        segm = np.zeros((B, V, 1, 24, 24), np.int64)
        for b in range(B):
            for v in range(V):
                yaw = yaw0[b] + 0.35 * t + v * math.pi / 2
                R = look_rotation(yaw, -0.1 + 0.05 * v)
                T = np.array([0.6 * math.cos(0.5 * t + b), 0.6 * math.sin(0.5 * t + b), 0.1 * b])
                d = room_depth(K, R, T, hw)
                raw = np.floor(d * 1000.0).astype(np.float32)
                raw[rng.random(raw.shape) < 0.01] = 0                       # sensor holes -> image maximum (PRE-FF:82)
                E = np.eye(4)
                E[:3, :3], E[:3, 3] = R.T, -R.T @ T
                out["depth_raw"][b, v], out["depth_m"][b, v] = raw, d.astype(np.float32)
                out["intrinsics"][b, v], out["rot"][b, v], out["trans"][b, v, :, 0], out["extrinsic"][b, v] = K, R, T, E
                blocks = rng.permutation(16).reshape(4, 4)
                segm[b, v, 0] = np.kron(blocks, np.ones((6, 6), np.int64))
        out["patch_segm"] = segm.reshape(B * V, 1, 24, 24)
        yield out
