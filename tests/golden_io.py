"""Shared helpers for golden fixtures: seed -> inputs regeneration and ragged packing.
The FF boundary is pinned on already-preprocessed inputs: depth24 / depth_full in
metres, CLIP grid features, dense patch_segm, habitat pose."""
from __future__ import annotations

import os

import numpy as np

from dynam3d_amd.synthetic import SyntheticEpisodes
from oracle import geometry as G

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJ_CASES = {
    # name: episode generator settings.  views > 1 = panorama (view ix looks along heading - ix*pi/6, VLN-FF:529-550);
    # pop = (step, env): `Feature_Fields.pop(env)` is called BEFORE that step (an episode ended, VLN-TR:778-784) and the
    # remaining environments carry on with their own streams.
    "walk": dict(B=2, steps=7, seed=1, grid_seed=5, stationary=False, wall=None, depth_hw=224),
    "wall": dict(B=2, steps=5, seed=2, grid_seed=6, stationary=True, wall=2.0, depth_hw=64),
    "pano3": dict(B=2, steps=4, seed=3, grid_seed=7, stationary=False, wall=None, depth_hw=64, views=3),
    "pop": dict(B=3, steps=5, seed=4, grid_seed=8, stationary=False, wall=None, depth_hw=64, pop=(3, 1)),
    # Pretrain class (PRE-FF) in inference mode: the four 90-degree views of `Net_3DFF.forward` (view_ids 0,3,6,9: view ix
    # looks along heading - view_ids[ix]*pi/6 in BOTH the cull and the unprojection, PRE-FF:696,920), 4 merge proposals.
    "prepano": dict(B=2, steps=3, seed=5, grid_seed=9, stationary=False, wall=None, depth_hw=64, views=4, view_ids=[0, 3, 6, 9],
                    variant="pretrain"),
    # Long horizon: 32 steps at bench-like state (>= 250 live instances per environment).  Random-depth steps delete what the frustum
    # re-observes closer (VLN-FF:329-396), the wall steps (a flat wall 2 m ahead) wipe most of the frustum at once: instance ids and
    # patch ids are recycled (VLN-FF:433-475), zone snapshots go stale and zones die and are re-opened (VLN-FF:694-756) for 30 steps
    # on top of each other.  `light`: only the key steps keep the large arrays; every step keeps the id orders, positions, row sums
    # and a hash of the integer bookkeeping.
    # Degenerate segmentations (the domain's edge cases): step 0 = ONE segment of all 576 patches (what `get_patch_segm` returns when FastSAM
    # fails, VLN-FF:424-430: the largest possible set, 577 encoder tokens), step 1 = 576 segments of one patch each (the most proposals /
    # merges / new instances a frame can produce), step 2 = a 1-patch segment beside a 575-patch one, step 3 = the usual 16 blocks, step 4 =
    # one segment again on top of a populated memory.  B = 1 (a batch of one is an edge case of every per-environment table too).
    "edge": dict(B=1, steps=5, seed=41, grid_seed=42, stationary=False, wall=None, depth_hw=64, segm_edge=True),
    "long": dict(B=2, steps=32, seed=21, grid_seed=22, stationary=False, wall=2.0, wall_steps=(9, 10, 19, 27), depth_hw=64,
                 light=True, key_steps=(0, 8, 9, 10, 11, 19, 20, 27, 28, 31)),
}


def int_hash(*arrays) -> np.int64:
    """64-bit digest of a sequence of integer arrays (shape-sensitive): the light form of a bookkeeping comparison."""
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(np.asarray(a, np.int64))
        h.update(np.asarray(a.shape, np.int64).tobytes())
        h.update(a.tobytes())
    return np.frombuffer(h.digest()[:8], np.int64)[0]


def make_gt(n_points, seed=0):
    """A synthetic ground-truth instance point cloud (Pretrain `gt_pcd_xyz` / `gt_pcd_label`): uniform points, instance id = the 1.5 m
    cell they fall into."""
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-9, 9, n_points), rng.uniform(-9, 9, n_points), rng.uniform(-3, 4, n_points)], 1).astype(np.float32)
    lab = (np.floor(xyz[:, 0] / 1.5).astype(np.int64) + 8) * 64 + (np.floor(xyz[:, 1] / 1.5).astype(np.int64) + 8)
    return xyz, lab


# Golden g21 (tests/golden/gen_golden_train.py): the reference's `update_feature_fields(is_training=True)` on the Pretrain panorama case
TRAIN_CASE = dict(traj="prepano", n_gt=20000, gt_seed=3, img_seed=77, steps=3)


def train_inputs(tc=TRAIN_CASE):
    """-> (case, gts [(xyz, label) per env], generator of (step inputs, image_ft (B, V, 768) float32))."""
    case = dict(TRAJ_CASES[tc["traj"]], steps=tc["steps"])
    gts = [make_gt(tc["n_gt"], seed=tc["gt_seed"] + b) for b in range(case["B"])]
    rng = np.random.default_rng(tc["img_seed"])

    def gen():
        for inp in traj_inputs(case):
            yield inp, rng.standard_normal((case["B"], len(case["view_ids"]), 768)).astype(np.float32)
    return case, gts, gen()


def traj_inputs(case):
    """Yields one dict per step for the environments still alive (see `pop`): depth_full (B,V,H,W), depth24 (B,V,576),
    grid (B,V,576,768), patch_segm (B*V,1,24,24) environment-major, positions, headings."""
    B0, V = case["B"], case.get("views", 1)
    eps = [SyntheticEpisodes(B0, seed=case["seed"] + 100 * v, stationary=case["stationary"], wall=case["wall"],
                             depth_hw=case["depth_hw"], image_hw=32, wall_steps=case.get("wall_steps")) for v in range(V)]
    rng = np.random.default_rng(case["grid_seed"])
    alive = list(range(B0))
    for t in range(case["steps"]):
        if case.get("pop") and case["pop"][0] == t:
            alive.pop(case["pop"][1])
        frs = [ep.next() for ep in eps]
        grid_all = rng.standard_normal((B0, V, 576, 768)).astype(np.float32)
        idx = np.asarray(alive)
        dfull = np.stack([G.preprocess_depth(fr.depth)[..., 0] for fr in frs], 1)[idx]                      # (B,V,H,W)
        d24 = np.stack([G.preprocess_depth(G.downsample_depth_nearest(fr.depth)).reshape(B0, 576) for fr in frs], 1)[idx]
        segm = np.stack([fr.patch_segm for fr in frs], 1)[idx]                                              # (B,V,1,24,24)
        if case.get("segm_edge"):
            flat = np.zeros((576,), np.int64)
            if t % 5 == 1:
                flat = np.arange(576, dtype=np.int64)
            elif t % 5 == 2:
                flat[300] = 1                                  # labels stay dense and in `torch.unique` order: {0: 575 patches, 1: one patch}
            elif t % 5 == 3:
                flat = segm[0, 0, 0].reshape(-1).copy()
            segm = np.broadcast_to(flat.reshape(1, 1, 1, 24, 24), segm.shape).copy()
        fr = frs[0]
        yield dict(depth_full=dfull.copy(), depth24=d24.copy(), grid=grid_all[idx].copy(),
                   patch_segm=segm.reshape(len(alive) * V, *segm.shape[2:]).copy(), positions=[fr.positions[i].tolist() for i in alive],
                   headings=[fr.headings[i] for i in alive], depth_raw=fr.depth[idx], rgb=fr.rgb[idx], alive=list(alive))


def pack_ragged(arrs):
    off = np.zeros(len(arrs) + 1, np.int64)
    for i, a in enumerate(arrs):
        off[i + 1] = off[i] + len(a)
    flat = np.concatenate([np.asarray(a, np.int64) for a in arrs]) if arrs else np.zeros(0, np.int64)
    return flat, off


def unpack_ragged(flat, off):
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


RENDER_CASES = {
    # name: synthetic scene + view settings for the Pretrain renderer goldens (g6)
    "small": dict(seed=21, n_points=576 * 2, H=6, W=6, n_samples=101, position=[0.2, 0.0, -0.4], heading=0.6),
    "full": dict(seed=22, n_points=576 * 6, H=12, W=12, n_samples=501, position=[0.3, 0.1, -0.2], heading=1.1),
}


def render_scene(case):
    """Stored patches of one environment: positions on a ring of surfaces around the origin (+ tomb-stones),
    directions, scales and fp16 features -- regenerated from the seed."""
    import math
    rng = np.random.default_rng(case["seed"])
    N = case["n_points"]
    ang, rad = rng.uniform(0, 2 * math.pi, N), rng.uniform(0.8, 3.5, N)
    pos = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(-1.2, 1.2, N)], -1).astype(np.float32)
    pos[::37] = -10000.0
    pdir = rng.uniform(0, 2 * math.pi, N).astype(np.float32)
    psc = rng.uniform(0.01, 0.3, N).astype(np.float32)
    fts = rng.standard_normal((N, 768)).astype(np.float16)
    return pos, pdir, psc, fts
