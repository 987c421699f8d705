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
