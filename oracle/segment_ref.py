"""CPU restatement (numpy float32, explicit operation order) of the on-device segmenter `d3d_segment_slic`.
TEST INFRASTRUCTURE (see oracle/geometry.py).

The reference's segmenter is FastSAM (VLN-FF:400-430), a network whose weights are not available offline: "parity unpinned" for the
masks themselves.  What the memory update consumes is the dense 24 x 24 label map built FROM masks (oracle/geometry.py::patch_segm, pinned by
g10); this file only defines the stand-in mask generator so that its HIP kernel can be checked bit for bit:
grid-seeded k-means over (r, g, b, x, y) with integer cluster sums, `iters` Lloyd updates, a final assignment, ties to the lowest index."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def segment_slic(rgb: np.ndarray, gx: int, gy: int, iters: int, compactness: float):
    """rgb (H, W, 3) uint8 -> (labels (H, W) int32, masks (gx*gy, H, W) uint8)."""
    H, W, _ = rgb.shape
    K = gx * gy
    S = F32(0.5) * (F32(W) / F32(gx) + F32(H) / F32(gy))
    w_xy = F32(F32(compactness) * F32(compactness)) / F32(S * S)
    cen = np.zeros((K, 5), F32)
    for k in range(K):
        i, j = divmod(k, gx)
        sx = F32(F32(j) + F32(0.5)) * F32(W) / F32(gx)
        sy = F32(F32(i) + F32(0.5)) * F32(H) / F32(gy)
        px, py = min(W - 1, int(sx)), min(H - 1, int(sy))
        cen[k] = [rgb[py, px, 0], rgb[py, px, 1], rgb[py, px, 2], sx, sy]
    pix = rgb.reshape(-1, 3).astype(F32)
    ys, xs = np.divmod(np.arange(H * W), W)
    fx, fy = xs.astype(F32) + F32(0.5), ys.astype(F32) + F32(0.5)
    lab = None
    for it in range(iters + 1):
        d = np.empty((H * W, K), F32)
        for k in range(K):
            dr, dg, db = pix[:, 0] - cen[k, 0], pix[:, 1] - cen[k, 1], pix[:, 2] - cen[k, 2]
            dx, dy = fx - cen[k, 3], fy - cen[k, 4]
            d[:, k] = ((dr * dr + dg * dg) + db * db) + w_xy * (dx * dx + dy * dy)
        lab = d.argmin(1)                                  # first minimum = lowest index on ties
        if it == iters:
            break
        for k in range(K):
            m = lab == k
            n = int(m.sum())
            if n:
                s = rgb.reshape(-1, 3)[m].astype(np.int64).sum(0)
                cen[k, :3] = s.astype(F32) / F32(n)
                cen[k, 3] = F32(xs[m].sum()) / F32(n) + F32(0.5)
                cen[k, 4] = F32(ys[m].sum()) / F32(n) + F32(0.5)
    lab = lab.reshape(H, W).astype(np.int32)
    masks = (lab[None] == np.arange(K)[:, None, None]).astype(np.uint8)
    return lab, masks
