"""Synthetic inputs of the Pretrain update for tools/bench_pretrain.py (no oracle import: depth preparation in numpy here)."""
import numpy as np


def _prep(depth, lo=0.0, hi=10.0):
    """preprocess_depth (VLN-POL:171-186): zero pixels <- column max, then metres."""
    d = depth.astype(np.float32).copy()
    mx = d.max(axis=1, keepdims=True)
    mx = np.broadcast_to(mx, d.shape)
    z = d == 0
    d[z] = mx[z]
    return (lo * 100.0 + d * (hi - lo) * 100.0) / 100.0


def step_inputs(eps, rng, B, V):
    frs = [ep.next() for ep in eps]
    H = frs[0].depth.shape[1]
    idx = np.minimum(np.floor(np.arange(24) * (H / 24)).astype(np.int64), H - 1)
    dfull = np.stack([_prep(fr.depth)[..., 0] for fr in frs], 1)
    d24 = np.stack([_prep(fr.depth[:, idx][:, :, idx]).reshape(B, 576) for fr in frs], 1)
    segm = np.stack([fr.patch_segm for fr in frs], 1).reshape(B * V, 1, 24, 24)
    grid = rng.standard_normal((B, V, 576, 768)).astype(np.float32)
    return dict(depth_full=dfull, depth24=d24, grid=grid, patch_segm=segm, positions=[p.tolist() for p in frs[0].positions], headings=list(frs[0].headings),
                img=rng.standard_normal((B, V, 768)).astype(np.float32))
