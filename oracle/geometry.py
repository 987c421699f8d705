"""CPU restatement (numpy, float32, explicit operation order) of the geometric operators on the
Dynam3D per-step path.  TEST INFRASTRUCTURE: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; the product never does.

Every function cites the reference lines it restates.  All arithmetic is float32 with one
rounding per operation (no FMA contraction) so the HIP kernels -- compiled with
-ffp-contract=off and written in the same operation order -- are BIT-EXACT against it.

Pinning: checked against the reference's own functions (imported in the build container by
oracle/ref_harness.py) and against the committed golden vectors tests/golden/g1..g3 generated
from them (tests/golden/gen_golden.py).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
TOMBSTONE = F32(-10000.0)  # VLN-FF:357


# ---------------------------------------------------------------------------------------------
# a1  Dynam3D_VLN.preprocess_depth                                   VLN-POL:171-186
# ---------------------------------------------------------------------------------------------
def preprocess_depth(depth: np.ndarray, depth_scale=(0.0, 10.0)) -> np.ndarray:
    """depth (B,H,W,1) f32 in [0,1].  Zero pixels take their COLUMN maximum (max over H), then
    d -> (min*100 + d*(max-min)*100)/100 in float32, operation order as the reference."""
    d = depth.astype(F32) * F32(1.0)
    col_max = d.max(axis=1, keepdims=True)
    d = np.where(d == 0, np.broadcast_to(col_max, d.shape), d).astype(F32)
    lo, hi = depth_scale
    t = (d * F32(hi - lo)).astype(F32)          # depth * (max-min)   (python float -> f32 scalar)
    t = (t * F32(100.0)).astype(F32)            # ... * 100.0
    t = (F32(lo * 100.0) + t).astype(F32)       # min*100 + ...
    return (t / F32(100.0)).astype(F32)


# ---------------------------------------------------------------------------------------------
# a2  cv2.resize(..., INTER_NEAREST)                                   VLN-POL:336-339
# ---------------------------------------------------------------------------------------------
def nearest_indices(src: int, dst: int) -> np.ndarray:
    """OpenCV INTER_NEAREST source index: min(floor(i * (src/dst)), src-1) with the scale in
    double precision (cv::resize computes inv_scale = 1./ (dst/src) in double)."""
    idx = np.floor(np.arange(dst, dtype=np.float64) * (float(src) / float(dst))).astype(np.int64)
    return np.minimum(idx, src - 1)


def downsample_depth_nearest(depth: np.ndarray, out_hw=(24, 24)) -> np.ndarray:
    """depth (B,H,W,1) -> (B,h,w,1).  'fixed' semantics of SURVEY F9 (per-image resize)."""
    B, H, W, _ = depth.shape
    ri, ci = nearest_indices(H, out_hw[0]), nearest_indices(W, out_hw[1])
    return depth[:, ri][:, :, ci]


# ---------------------------------------------------------------------------------------------
# camera tables shared by a5 / a13                                     VLN-FF:283-287, 307-311
# ---------------------------------------------------------------------------------------------
def camera_tables(H: int = 24, W: int = 24, hfov: float = 90.0, vfov: float = 90.0):
    """Per-patch tangent tables in row-major patch order p = r*W + c.
    tan_xy[p] = f32(c_off/halfW + 1/W) * tan(pi*hfov/360)   with c_off = c - W//2
    tan_z [p] = f32(r_off/halfH - 1/H) * tan(pi*vfov/360)   with r_off = H//2 - r
    dir0  [p] = -arctan(tan_xy[p])            (float32 arctan)
    scale_k   = tan(pi*hfov/360) * 2 / W  applied as ((d * tan) * 2.) / W in f32."""
    hW, hH = W // 2, H // 2
    th, tv = math.tan(math.pi * hfov / 360.0), math.tan(math.pi * vfov / 360.0)
    row_xy = np.array([i / hW + 1 / W for i in range(-hW, hW)], F32)
    tan_xy = (np.tile(row_xy, H) * F32(th)).astype(F32)
    col_z = np.array([i / hH - 1 / H for i in range(hH, -hH, -1)], F32)
    tan_z = (np.repeat(col_z, W) * F32(tv)).astype(F32)
    dir0 = (-np.arctan(tan_xy)).astype(F32)
    return tan_xy, tan_z, dir0, F32(th)


TWO_PI_F32 = F32(2 * math.pi)


def _pymod(a: np.ndarray, b: np.float32) -> np.ndarray:
    """numpy float32 `%` (npy_divmodf): fmod, then add the divisor when the signs differ."""
    m = np.fmod(a, b).astype(F32)
    fix = (m != 0) & ((m < 0) != (b < 0))
    return np.where(fix, (m + b).astype(F32), m).astype(F32)


# ---------------------------------------------------------------------------------------------
# a5  project_depth_to_3d_habitat (+ world offset)                      VLN-FF:276-293, 550-554
# ---------------------------------------------------------------------------------------------
def unproject_habitat(depth24: np.ndarray, position_habitat, heading: float, H=24, W=24, hfov=90.0, vfov=90.0):
    """depth24 (P,) f32 metres; position habitat (x,y,z); heading rad (view heading already
    includes the -pi/6*view offset, VLN-FF:550).
    Returns pos (P,3) world f32, direction (P,) f32 in [0,2pi), scale (P,) f32."""
    tan_xy, tan_z, dir0, th = camera_tables(H, W, hfov, vfov)
    d = depth24.astype(F32).reshape(-1)
    c, s = F32(math.cos(heading)), F32(math.sin(heading))
    dx = (d * tan_xy).astype(F32)
    dz = (d * tan_z).astype(F32)
    scale = (((d * th).astype(F32) * F32(2.0)).astype(F32) / F32(W)).astype(F32)
    direction = _pymod((dir0 + F32(heading)).astype(F32), TWO_PI_F32)
    rel_x = ((dx * c).astype(F32) - (d * s).astype(F32)).astype(F32)
    rel_y = ((dx * s).astype(F32) + (d * c).astype(F32)).astype(F32)
    # world = rel + (px, -pz, py)   (axis swap VLN-FF:523)
    wx, wy, wz = F32(position_habitat[0]), F32(-position_habitat[2]), F32(position_habitat[1])
    pos = np.stack([(rel_x + wx).astype(F32), (rel_y + wy).astype(F32), (dz + wz).astype(F32)], axis=-1)
    return pos, direction, scale


# ---------------------------------------------------------------------------------------------
# a13 get_patch_3d_info                                                 VLN-FF:296-326
# ---------------------------------------------------------------------------------------------
def patch_3d_info(depth24: np.ndarray, H=24, W=24, hfov=90.0, vfov=90.0):
    """depth24 (N,P) -> rel_x, rel_y, rel_z, direction, scale each (N,P,1) f32 (camera frame)."""
    tan_xy, tan_z, dir0, th = camera_tables(H, W, hfov, vfov)
    d = depth24.astype(F32)
    rel_x = (d * tan_xy[None]).astype(F32)
    rel_z = (d * tan_z[None]).astype(F32)
    scale = (((d * th).astype(F32) * F32(2.0)).astype(F32) / F32(W)).astype(F32)
    direction = np.broadcast_to(_pymod(dir0, TWO_PI_F32)[None], d.shape).astype(F32)
    e = lambda a: a[..., None]
    return e(rel_x), e(d), e(rel_z), e(direction), e(scale)


# ---------------------------------------------------------------------------------------------
# a4  get_frustum_mask_habitat + depth test                             VLN-FF:88-115, 349-353
# ---------------------------------------------------------------------------------------------
def frustum_mask_habitat(points: np.ndarray, depth_img: np.ndarray, position_habitat, heading: float,
                         hfov=90.0, vfov=90.0, near=0.0, far=3.0, slack=0.1):
    """points (N,3) world f32; depth_img (Hd,Wd) f32 metres.  Returns bool mask (N,) of stored
    points that the current view re-observes (to be tomb-stoned).

    Projection (VLN-FF:106, torch.einsum with K = [[fx,0,cx],[0,fy,cy],[0,0,1]]):
        u_h = fx*X + cx*Z   v_h = fy*Y + cy*Z   (each product rounded, then one add; the zero
        terms contribute exact zeros) ; u = trunc(u_h / Z), v = trunc(v_h / Z).
    Non-finite or out-of-int64 quotients are 'outside' (x86 cvttss2si -> INT64_MIN)."""
    Hd, Wd = depth_img.shape
    fx = F32(Wd / np.tan(np.deg2rad(hfov) / 2.0) / 2.0)
    fy = F32(Hd / np.tan(np.deg2rad(vfov) / 2.0) / 2.0)
    cx, cy = F32(Wd / 2.0), F32(Hd / 2.0)
    cam = (F32(position_habitat[0]), F32(-position_habitat[2]), F32(position_habitat[1]))
    a = -heading
    c, s = F32(math.cos(a)), F32(math.sin(a))
    p = points.astype(F32)
    px = (p[:, 0] - cam[0]).astype(F32)
    py = (p[:, 1] - cam[1]).astype(F32)
    pz = (p[:, 2] - cam[2]).astype(F32)
    rx = ((px * c).astype(F32) - (py * s).astype(F32)).astype(F32)
    ry = ((px * s).astype(F32) + (py * c).astype(F32)).astype(F32)
    X, Y, Z = rx, (-pz).astype(F32), ry            # (rel_x, -rel_z, rel_y)  VLN-FF:102
    with np.errstate(all="ignore"):
        uh = ((fx * X).astype(F32) + (cx * Z).astype(F32)).astype(F32)
        vh = ((fy * Y).astype(F32) + (cy * Z).astype(F32)).astype(F32)
        uf = (uh / Z).astype(F32)
        vf = (vh / Z).astype(F32)
    ok = np.isfinite(uf) & np.isfinite(vf) & (np.abs(uf) < F32(2.0 ** 62)) & (np.abs(vf) < F32(2.0 ** 62))
    u = np.where(ok, np.trunc(np.where(ok, uf, 0)), -1).astype(np.int64)
    v = np.where(ok, np.trunc(np.where(ok, vf, 0)), -1).astype(np.int64)
    inside = ok & (Z >= F32(near)) & (Z <= F32(far)) & (u >= 0) & (u <= Wd - 1) & (v >= 0) & (v <= Hd - 1)
    uu, vv = np.clip(u, 0, Wd - 1), np.clip(v, 0, Hd - 1)
    cam_d = depth_img.astype(F32)[vv, uu]
    return inside & (Z < (cam_d + F32(slack)).astype(F32))


# ---------------------------------------------------------------------------------------------
# a8  torch_kdtree build/query (third-party, absent; parity defined)     VLN-FF:246, 606-610
# ---------------------------------------------------------------------------------------------
def knn_bruteforce(points: np.ndarray, queries: np.ndarray, k: int):
    """Ascending (dist^2, index) k nearest neighbours; d2 = ((dx*dx + dy*dy) + dz*dz) in f32,
    ties -> lowest index.  Returns (d2 (M,k) f32, idx (M,k) int64)."""
    p, q = points.astype(F32), queries.astype(F32)
    M = q.shape[0]
    if k == 0 or p.shape[0] == 0:
        return np.zeros((M, 0), F32), np.zeros((M, 0), np.int64)
    k = min(k, p.shape[0])
    out_d, out_i = np.empty((M, k), F32), np.empty((M, k), np.int64)
    chunk = max(1, (1 << 24) // max(p.shape[0], 1))
    for a in range(0, M, chunk):
        qq = q[a:a + chunk]
        with np.errstate(over="ignore"):
            dx = (qq[:, None, 0] - p[None, :, 0]).astype(F32)
            dy = (qq[:, None, 1] - p[None, :, 1]).astype(F32)
            dz = (qq[:, None, 2] - p[None, :, 2]).astype(F32)
            d2 = (((dx * dx).astype(F32) + (dy * dy).astype(F32)).astype(F32) + (dz * dz).astype(F32)).astype(F32)
        rows = np.arange(d2.shape[0])
        for j in range(k):                      # k passes of argmin: first occurrence == lowest index on ties
            i = np.argmin(d2, axis=1)
            out_i[a:a + chunk, j] = i
            out_d[a:a + chunk, j] = d2[rows, i]
            d2[rows, i] = np.inf
    return out_d, out_i


# ---------------------------------------------------------------------------------------------
# a7 (geometry part)  per-segment centroid + 7-vector                    VLN-FF:582-591
# ---------------------------------------------------------------------------------------------
def mean_rows_f64(x: np.ndarray) -> np.ndarray:
    """Defined reduction: sequential float64 sum over rows, divide, round once to float32.
    (torch's float32 mean uses a machine-dependent vectorised order; this is within 1 ulp of it.)"""
    acc = np.zeros(x.shape[1:], np.float64)
    for r in x.astype(np.float64):
        acc = acc + r
    return (acc / x.shape[0]).astype(F32)


def segment_geometry(pos: np.ndarray, direction: np.ndarray, scale: np.ndarray, centroid: np.ndarray) -> np.ndarray:
    """7-vector [pos - centroid (3), ||pos|| (1), sin dir, cos dir, scale]  (VLN-FF:584-591).
    ||pos|| = sqrt((x*x + y*y) + z*z) in f32; sin/cos are float32 library calls (<=1 ulp class,
    compared with tolerance, never bit-exact)."""
    p = pos.astype(F32)
    rel = (p - centroid[None].astype(F32)).astype(F32)
    n2 = (((p[:, 0] * p[:, 0]).astype(F32) + (p[:, 1] * p[:, 1]).astype(F32)).astype(F32) + (p[:, 2] * p[:, 2]).astype(F32)).astype(F32)
    dist = np.sqrt(n2).astype(F32)
    return np.concatenate([rel, dist[:, None], np.sin(direction.astype(F32))[:, None].astype(F32),
                           np.cos(direction.astype(F32))[:, None].astype(F32), scale.astype(F32)[:, None]], axis=1)


# ---------------------------------------------------------------------------------------------
# a11 zone cell centre                                                   VLN-FF:694-695
# ---------------------------------------------------------------------------------------------
def zone_cell_centre(pos: np.ndarray, cell=(2.0, 2.0, 2.0)) -> np.ndarray:
    p = pos.astype(F32)
    out = np.empty_like(p)
    for a in range(3):
        L = F32(cell[a])
        out[:, a] = ((np.floor((p[:, a] / L).astype(F32)) * L).astype(F32) + F32(cell[a] / 2.0)).astype(F32)
    return out


# ---------------------------------------------------------------------------------------------
# a12 agent-frame transform + radius filter                               VLN-FF:829-841
# ---------------------------------------------------------------------------------------------
def agent_frame(pos: np.ndarray, position_habitat, heading: float, radius: float):
    cam = (F32(position_habitat[0]), F32(-position_habitat[2]), F32(position_habitat[1]))
    a = -heading
    c, s = F32(math.cos(a)), F32(math.sin(a))
    p = pos.astype(F32).reshape(-1, 3)
    px = (p[:, 0] - cam[0]).astype(F32)
    py = (p[:, 1] - cam[1]).astype(F32)
    pz = (p[:, 2] - cam[2]).astype(F32)
    rx = ((px * c).astype(F32) - (py * s).astype(F32)).astype(F32)
    ry = ((px * s).astype(F32) + (py * c).astype(F32)).astype(F32)
    rel = np.stack([rx, ry, pz], axis=-1)
    with np.errstate(over="ignore"):
        n2 = (((rx * rx).astype(F32) + (ry * ry).astype(F32)).astype(F32) + (pz * pz).astype(F32)).astype(F32)
    keep = np.sqrt(n2).astype(F32) <= F32(radius)
    return rel, keep


# =============================================================================================
# Intrinsics / extrinsics path ("most 3D datasets": posed RGB-D with a pinhole camera)      SURVEY.md 8f-2
# =============================================================================================
# ---------------------------------------------------------------------------------------------
# get_frustum_mask + depth test                                          PRE-FF:98-118, 693-704
# ---------------------------------------------------------------------------------------------
def frustum_mask_pinhole(points: np.ndarray, depth_img: np.ndarray, intrinsics: np.ndarray, view_matrix: np.ndarray,
                         near=0.0, far=3.0, slack=0.1):
    """points (N,3) world f32; depth_img (H,W) f32 metres; intrinsics (>=3,>=3), view_matrix (4,4) world->camera, f32.
    `einsum("b c, N c -> N b")` rows: view = fma(V3,1, fma(V2,z, fma(V1,y, V0*x))), uv = fma(K2,Z, fma(K1,Y, K0*X)) -- the
    accumulation ATen's CPU matmul performs for these shapes (found by matching all 36 000 intermediate values of the
    golden cases bit for bit) and the one this build's kernel uses (`fmaf`); pinned by tests/golden/g2b_frustum_pinhole.npz,
    generated by the reference function.  u, v = trunc(uv / uv_z) to int64,
    `depth` = camera-frame Z; inside = near <= Z <= far, 0 <= u <= W-1, 0 <= v <= H-1; hit = inside & Z < depth[v,u] + slack."""
    H, W = depth_img.shape
    K = np.asarray(intrinsics, F32)[:3, :3]
    V = np.asarray(view_matrix, F32)
    p = points.astype(F32)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]

    def fma(a, b, acc):                     # float32 fused multiply-add: the 24x24-bit product is exact in float64
        return (np.float64(a) * b.astype(np.float64) + acc.astype(np.float64)).astype(F32)

    def row(m, a, b, c, last=None):         # ATen's CPU matmul for these shapes: acc = m0*a, then one FMA per further term
        s = fma(m[2], c, fma(m[1], b, (m[0] * a).astype(F32)))
        return s if last is None else fma(last, np.ones_like(a), s)

    with np.errstate(all="ignore"):
        X, Y, Z = (row(V[r], x, y, z, V[r, 3]) for r in range(3))
        uh, vh, zh = (row(K[r], X, Y, Z) for r in range(3))
        uf, vf = (uh / zh).astype(F32), (vh / zh).astype(F32)
    ok = np.isfinite(uf) & np.isfinite(vf) & (np.abs(uf) < F32(2.0 ** 62)) & (np.abs(vf) < F32(2.0 ** 62))
    u = np.where(ok, np.trunc(np.where(ok, uf, 0)), -1).astype(np.int64)
    v = np.where(ok, np.trunc(np.where(ok, vf, 0)), -1).astype(np.int64)
    inside = ok & (Z >= F32(near)) & (Z <= F32(far)) & (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= H - 1)
    cam_d = depth_img.astype(F32)[np.clip(v, 0, H - 1), np.clip(u, 0, W - 1)]
    return inside & (Z < (cam_d + F32(slack)).astype(F32))


# ---------------------------------------------------------------------------------------------
# project_depth_to_3d (Open3D create_from_depth_image + nearest resize)          PRE-FF:81-94
# ---------------------------------------------------------------------------------------------
def torch_nearest_indices(src: int, dst: int) -> np.ndarray:
    """F.interpolate(mode='nearest') source index: min(int(floorf(i * scale)), src-1), scale = float32(src)/dst
    (ATen `nearest_idx`, float32 arithmetic)."""
    scale = F32(src) / F32(dst)
    return np.minimum(np.floor((np.arange(dst, dtype=F32) * scale).astype(F32)).astype(np.int64), src - 1)


def project_depth_to_3d(depth: np.ndarray, intrinsic: np.ndarray, depth_scale=1000.0, depth_trunc=1000.0, out_hw=(24, 24)):
    """depth (H,W) raw sensor units (the reference casts to uint16) -> (points (h*w,3) float64 camera frame, mask).

    PARITY UNPINNED for the Open3D call (open3d==0.18 is not installed; no reference test holds a vector).  Restated from
    Open3D's published algorithm (PointCloudFactory.cpp, `CreatePointCloudFromFloatDepthImage` after
    `ConvertDepthToFloatImage`): d = float(uint16) / float(depth_scale), d >= depth_trunc -> 0; pixel (row i, col j) with
    d > 0 gives z = d, x = (j - cx) * z / fx, y = (i - cy) * z / fy in double; INVALID PIXELS ARE DROPPED, so the reference's
    `.view(H, W, 3)` raises whenever one exists and its `except` returns all-zero points (PRE-FF:90-91) -- restated as such.
    Zero pixels are first replaced by the image maximum (PRE-FF:82)."""
    d = np.array(depth, dtype=np.float64)
    d[d == 0] = d.max() if d.size else 0
    u16 = d.astype(np.uint16)
    f = (u16.astype(F32) / F32(depth_scale)).astype(F32)
    f[f >= F32(depth_trunc)] = 0
    H, W = f.shape
    h, w = out_hw
    if not np.all(f > 0):
        pts = np.zeros((h * w, 3), np.float64)
    else:
        fx, fy, cx, cy = float(intrinsic[0][0]), float(intrinsic[1][1]), float(intrinsic[0][2]), float(intrinsic[1][2])
        ri, ci = torch_nearest_indices(H, h), torch_nearest_indices(W, w)
        z = f[np.ix_(ri, ci)].astype(np.float64)
        x = (ci[None, :].astype(np.float64) - cx) * z / fx
        y = (ri[:, None].astype(np.float64) - cy) * z / fy
        pts = np.stack([x, y, z], -1).reshape(-1, 3)
    return pts, pts[:, 2] > 0.002


def heading_angle(position: np.ndarray) -> np.ndarray:
    """Feature_Fields.get_heading_angle (PRE-FF:378-387), dtype-preserving like the reference (float64 in, float64 out)."""
    dx, dy = position[:, 0], position[:, 1]
    xy = np.sqrt(np.square(dx) + np.square(dy))
    xy[xy < 1e-4] = 1e-4
    h = -np.arcsin(dx / xy)
    h[dy < 0] = h[dy < 0] - np.pi
    return h


def rays_pinhole(fx: float, fy: float, view_h=12, view_w=12, near=0.0, far=10.0, n_samples=501):
    """Feature_Fields.get_rays (PRE-FF:390-405): N constant-depth float32 images at near + spacing*(i+1) unprojected with
    PinholeCameraIntrinsic(view_w, view_h, fx, fy, view_w/2, view_h/2) (Open3D: double arithmetic on the float32 depth).
    -> rel_position (R,N,3) f64, rel_direction (R,1) f64 = -arctan(x/z) of the last sample, rel_dist (R,N) f64."""
    spacing = (far - near) / n_samples
    z = np.array([F32(near + spacing * (i + 1)) for i in range(n_samples)], np.float64)       # np.full(..., dtype=float32)
    jj, ii = np.meshgrid(np.arange(view_w, dtype=np.float64), np.arange(view_h, dtype=np.float64))
    cx, cy = view_w / 2, view_h / 2
    x = (jj.reshape(-1, 1) - cx) * z[None, :] / fx
    y = (ii.reshape(-1, 1) - cy) * z[None, :] / fy
    rel = np.stack([x, y, np.broadcast_to(z[None, :], x.shape)], -1)
    return rel, -np.arctan(rel[:, -1:, 0] / rel[:, -1:, 2]), rel[..., 2].copy()


def unproject_pinhole(depth: np.ndarray, intrinsic: np.ndarray, R: np.ndarray, T: np.ndarray, scale_tan: float, input_width=24,
                      depth_scale=1000.0, depth_trunc=1000.0, out_hw=(24, 24)):
    """Per-view patch geometry of the intrinsics branch of update_feature_fields (PRE-FF:905-916):
    scale = z32 * |tan(rel_direction[0][-1])| * 2 / input_width (float32 array ops), world = R @ p32 + T (float64, then
    float32), direction = get_heading_angle(world) (float64, then float32).  `scale_tan` = |tan(rel_direction[0][-1])| of the
    rays built from the view-sized intrinsics (PRE-FF:849-856)."""
    pts, _ = project_depth_to_3d(depth, intrinsic, depth_scale, depth_trunc, out_hw)
    p32 = pts.astype(F32)
    sc = (p32[:, -1] * F32(scale_tan)).astype(F32)
    sc = (sc * F32(2.0)).astype(F32)
    sc = (sc / F32(input_width)).astype(F32)
    world = (np.asarray(R, np.float64) @ p32.T.astype(np.float64) + np.asarray(T, np.float64).reshape(3, 1)).T
    return world.astype(F32), heading_angle(world).astype(F32), sc


def view_scale_tan(intrinsic: np.ndarray, depth_hw, view_hw=(12, 12)) -> float:
    """|tan(rel_direction[0][-1])| for the rays of `init_camera_intrinsic` (PRE-FF:849-856): fx scaled by view_w / depth_W,
    principal point view_w/2; ray 0 is pixel (0,0): rel_direction = -arctan(((0 - cx) * z / fx) / z)."""
    fx = float(intrinsic[0][0]) * (view_hw[1] / depth_hw[1])
    z = float(F32(10.0))
    x = (0.0 - view_hw[1] / 2) * z / fx
    return math.fabs(math.tan(float(-np.arctan(x / z))))


# ---------------------------------------------------------------------------------------------
# a6  get_patch_segm post-processing (everything after FastSAM)                   VLN-FF:407-420
# ---------------------------------------------------------------------------------------------
def patch_segm_from_masks(masks: np.ndarray, out_hw=(24, 24)) -> np.ndarray:
    """masks (n,H,W) in {0,1} (FastSAM 'everything' masks) -> dense labels (1,h,w) int64.
    patch_group starts as masks[0] and every mask g overwrites its pixels with g ('last mask wins'; uncovered pixels keep
    masks[0]'s 0, i.e. they join group 0); F.interpolate(mode='nearest') to (h,w) (ATen float32 index rule); labels replaced
    by their rank in torch.unique (sorted) order.  No masks -> the reference's `masks[0]` raises -> all zeros (VLN-FF:424-426)."""
    h, w = out_hw
    if masks.shape[0] == 0:
        return np.zeros((1, h, w), np.int64)
    group = masks[0].astype(np.float32).copy()
    for g in range(masks.shape[0]):
        group[masks[g] == 1] = g
    ri, ci = torch_nearest_indices(masks.shape[1], h), torch_nearest_indices(masks.shape[2], w)
    small = group[np.ix_(ri, ci)].astype(np.int64)
    out = small.copy()
    for rank, lab in enumerate(np.unique(small).tolist()):
        out[small == lab] = rank
    return out[None]
