"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE's own code.

Container-only (needs /root/reference; see oracle/ref_harness.py).  Run:
    python tests/golden/gen_golden.py [g1 g2 g3 g4 g5 g6 g7]
Only data (inputs/expected outputs) is written; no reference source travels.  Inputs that are
cheap to regenerate (synthetic episodes, random grid features, synthetic weights) are stored as
SEEDS -- `tests/golden_io.py` rebuilds them with the same generators.
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from tests.golden_io import traj_inputs, TRAJ_CASES, pack_ragged, int_hash  # noqa: E402
from dynam3d_amd.weights import ff_param_spec, synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def g1_unproject():
    m = rh.load_ref_module("vln")
    sys.argv = ["x"]
    F = m.Feature_Fields(batch_size=1, device="cpu")
    rng = np.random.default_rng(100)
    cases = {}
    heads = [0.0, 0.7, math.pi, 5.5, 2 * math.pi - 1e-7, 1e-9]
    for i, h in enumerate(heads):
        d = rng.uniform(0.05, 10.0, (1, 576)).astype(np.float32)
        if i == 1:
            d[0, :7] = 0.0
        pos = [float(x) for x in rng.uniform(-20, 20, 3)]
        rx, ry, rz, dr, sc = F.project_depth_to_3d_habitat(d, h)
        w = [pos[0], -pos[2], pos[1]]
        cases[f"depth_{i}"] = d[0]
        cases[f"heading_{i}"] = np.float64(h)
        cases[f"position_{i}"] = np.array(pos, np.float64)
        cases[f"pos_{i}"] = np.stack([(rx + w[0])[0], (ry + w[1])[0], (rz + w[2])[0]], -1)
        cases[f"dir_{i}"] = dr
        cases[f"scale_{i}"] = sc
    d = rng.uniform(0.05, 10.0, (3, 576)).astype(np.float32)
    info = F.get_patch_3d_info(d)
    cases["info_depth"] = d
    for n, t in zip(["rel_x", "rel_y", "rel_z", "direction", "scale"], info):
        cases["info_" + n] = t.numpy()
    cases["n"] = np.int64(len(heads))
    np.savez_compressed(os.path.join(OUT, "g1_unproject.npz"), **cases)
    print("g1 ok")


def g2_frustum():
    m = rh.load_ref_module("vln")
    rng = np.random.default_rng(200)
    cases = {}
    n = 0
    for Hd, N in [(64, 2000), (256, 2000), (224, 4000)]:
        pts = rng.uniform(-5, 5, (N, 3)).astype(np.float32)
        pts[:40] = -10000.0
        pos = [float(x) for x in rng.uniform(-1.5, 1.5, 3)]
        h = float(rng.uniform(0, 2 * math.pi))
        cam = np.array([pos[0], -pos[2], pos[1]], np.float32)
        pts[40:48] = cam                                   # z == 0 -> inf / nan quotients
        pts[48:56] = cam + np.array([1e-30, 0, 0], np.float32)
        # points marginally left of / above the image: negative fractional u, v (trunc -> 0)
        a = -h
        for j in range(16):
            z = 1.0 + 0.1 * j
            x_cam = -z * (1.0 + (j % 4 - 1) * 1e-3)        # u_h/z close to 0
            # invert (rx, -rz, ry) rotation: rel = R(-h) * (p - cam)
            rx, ry = x_cam, z
            px = rx * math.cos(a) + ry * math.sin(a)
            py = -rx * math.sin(a) + ry * math.cos(a)
            pts[56 + j] = cam + np.array([px, py, 0.0], np.float32)
        dimg = rng.uniform(0.3, 4.0, (Hd, Hd)).astype(np.float32)
        position = [pos[0], -pos[2], pos[1]]
        mask, depth, u, v = m.get_frustum_mask_habitat(torch.tensor(pts), Hd, Hd, 90.0, 90.0, position, h, far=3.0)
        uu, vv = u % Hd, v % Hd
        fm = (mask & (depth < torch.tensor(dimg)[vv, uu] + 0.1)).numpy()
        cases[f"pts_{n}"], cases[f"depth_{n}"] = pts, dimg
        cases[f"position_{n}"], cases[f"heading_{n}"] = np.array(pos, np.float64), np.float64(h)
        cases[f"mask_{n}"] = fm
        n += 1
    cases["n"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "g2_frustum.npz"), **cases)
    print("g2 ok", [int(cases[f'mask_{i}'].sum()) for i in range(n)])


def g2b_frustum_pinhole():
    """Intrinsics/extrinsics cull (PRE-FF:98-118 + the depth test of PRE-FF:699-704) and get_heading_angle (PRE-FF:378-387),
    both executed by the reference's Pretrain module."""
    m = rh.load_ref_module("pre")
    sys.argv = ["x"]
    F = m.Feature_Fields(batch_size=1, device="cpu")
    rng = np.random.default_rng(250)
    cases = {}
    n = 0
    for (Hd, Wd), N in [((48, 64), 3000), ((480, 640), 4000), ((240, 320), 3000)]:
        ang = rng.uniform(0, 2 * math.pi, 3)
        cz, sz = math.cos(ang[0]), math.sin(ang[0])
        cy_, sy = math.cos(ang[1] * 0.2), math.sin(ang[1] * 0.2)
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
        R = Rz @ Ry                                               # camera -> world
        T = rng.uniform(-1.5, 1.5, 3)
        view = np.eye(4)
        view[:3, :3], view[:3, 3] = R.T, -R.T @ T                 # world -> camera
        K = np.eye(4)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 0.9 * Wd, 0.9 * Wd, Wd / 2 - 0.5, Hd / 2 + 0.25
        cam_pts = np.stack([rng.uniform(-3, 3, N), rng.uniform(-3, 3, N), rng.uniform(-1, 5, N)], -1)
        pts = (R @ cam_pts.T + T[:, None]).T.astype(np.float32)
        pts[:40] = -10000.0
        pts[40:44] = T.astype(np.float32)                         # camera centre: z = 0 -> inf / nan quotients
        dimg = rng.uniform(0.3, 4.0, (Hd, Wd)).astype(np.float32)
        tp = torch.tensor(pts)
        mask, depth, u, v = m.get_frustum_mask(tp, Hd, Wd, torch.tensor(K, dtype=torch.float32), torch.tensor(view, dtype=torch.float32), far=3.0)
        uu, vv = u % Wd, v % Hd
        fm = (mask & (depth < torch.tensor(dimg)[vv, uu] + 0.1)).numpy()
        cases[f"pts_{n}"], cases[f"depth_{n}"], cases[f"K_{n}"], cases[f"view_{n}"] = pts, dimg, K.astype(np.float32), view.astype(np.float32)
        cases[f"mask_{n}"] = fm
        n += 1
    cases["n"] = np.int64(n)
    hp = rng.uniform(-4, 4, (400, 3))
    hp[:8, :2] *= 1e-6                                            # xy_dist < 1e-4 clamp
    hp[8:12, 1] = 0.0
    cases["heading_pts"] = hp
    cases["heading_out"] = F.get_heading_angle(hp.copy())
    np.savez_compressed(os.path.join(OUT, "g2b_frustum_pinhole.npz"), **cases)
    print("g2b ok", [int(cases[f'mask_{i}'].sum()) for i in range(n)])


def g3_knn():
    rng = np.random.default_rng(300)
    cases = {}
    n = 0
    for M in (1, 2, 17, 300):
        for nq in (1, 16, 40):
            for k in (1, 2, 4):
                if k > M:
                    continue
                pts = rng.uniform(-8, 8, (M, 3)).astype(np.float32)
                q = rng.uniform(-8, 8, (nq, 3)).astype(np.float32)
                if M >= 17:
                    pts[3] = pts[11]                       # exact duplicate -> tie
                    pts[5] = -10000.0                      # tomb-stoned slots
                    pts[6] = -10000.0
                    q[0] = pts[3]                          # zero distance + tie
                    pts[8] = np.round(pts[8])              # grid points -> more ties
                    pts[9] = pts[8] + np.array([1, 0, 0], np.float32)
                    pts[10] = pts[8] - np.array([1, 0, 0], np.float32)
                    if nq > 1:
                        q[1] = pts[8]
                tree = rh._BruteTree(torch.from_numpy(pts))
                d2, idx = tree.query(torch.from_numpy(q), nr_nns_searches=k)
                cases[f"pts_{n}"], cases[f"q_{n}"], cases[f"k_{n}"] = pts, q, np.int64(k)
                cases[f"d2_{n}"], cases[f"idx_{n}"] = d2.numpy(), idx.numpy()
                n += 1
    cases["n"] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, "g3_knn.npz"), **cases)
    print("g3 ok", n)


def g4_trajectories():
    """Full Feature_Fields trajectories from the REFERENCE class (VLN-FF), per step and env:
    bookkeeping dicts (exact), instance/zone stores, get_environment_features outputs."""
    sd = synth_state_dict(ff_param_spec(), seed=0)
    only = os.environ.get("G4_ONLY")
    for name, case in TRAJ_CASES.items():
        if only and name not in only.split(","):
            continue
        B, steps = case["B"], case["steps"]
        pre = case.get("variant") == "pretrain"
        if pre:                                                      # the Pretrain class also owns the renderer's parameters
            from dynam3d_amd.weights import render_param_spec
            ref = rh.RefFeatureFields(B, synth_state_dict(ff_param_spec() + render_param_spec(), seed=0), which="pre")
        else:
            ref = rh.RefFeatureFields(B, sd)
        out = {"B": np.int64(B), "steps": np.int64(steps)}
        V = case.get("views", 1)
        for t, inp in enumerate(traj_inputs(case)):
            if case.get("pop") and case["pop"][0] == t:
                ref.F.pop(case["pop"][1])                       # the reference's own pop (VLN-FF:219-243)
                B = ref.F.batch_size
            if pre:                                                  # PRE-FF has no get_environment_features: state only
                ref.step_pretrain(torch.from_numpy(inp["depth_full"]), inp["depth24"], inp["grid"], torch.from_numpy(inp["patch_segm"]),
                                  inp["positions"], inp["headings"], case["view_ids"])
                er = None
            else:
                er = ref.step(torch.from_numpy(inp["depth_full"]), inp["depth24"], inp["grid"], torch.from_numpy(inp["patch_segm"]),
                              inp["positions"], inp["headings"], num_of_views=V)
            F = ref.F
            out[f"t{t}_B"] = np.int64(B)
            light = bool(case.get("light")) and t not in case.get("key_steps", ())
            for b in range(B):
                p = f"t{t}_b{b}_"
                own = F.global_patch_to_instance_dict[b]
                ks = np.array(sorted(own.keys()), np.int64)
                mem = F.global_instance_to_patch_dict[b]
                zm = F.global_zone_to_instance_dict[b]
                owner_inst = np.array([own[k] for k in ks.tolist()], np.int64)
                im, imo = pack_ragged([mem[k] for k in mem])
                zmm, zmo = pack_ragged([zm[k] for k in zm])
                rows_pos = np.asarray(F.global_patch_position[b]).astype(np.float32)
                out[p + "inst_order"] = np.array(list(mem.keys()), np.int64)
                out[p + "zone_order"] = np.array(list(zm.keys()), np.int64)
                if light:          # (long trajectories) digests of the large integer tables and of the row store's bits
                    out[p + "book_hash"] = int_hash(ks, owner_inst, im, imo, zmm, zmo)
                    out[p + "rows_hash"] = int_hash(rows_pos.view(np.uint32))
                    out[p + "n_rows"] = np.int64(rows_pos.shape[0])
                else:
                    out[p + "owner_ids"], out[p + "owner_inst"] = ks, owner_inst
                    out[p + "inst_members"], out[p + "inst_members_off"] = im, imo
                    out[p + "zone_members"], out[p + "zone_members_off"] = zmm, zmo
                    out[p + "rows_pos"] = rows_pos
                zk = F.global_zone_key_to_id[b]
                out[p + "zone_keys"] = np.array(list(zk.keys()), np.float32).reshape(-1, 3)
                out[p + "zone_key_ids"] = np.array(list(zk.values()), np.int64)
                ip, iF = F.global_instance_position[b].numpy(), F.global_instance_fts[b].numpy()
                zp, zF = F.global_zone_position[b].numpy(), F.global_zone_fts[b].numpy()
                out[p + "ipos"], out[p + "zpos"] = ip.copy(), zp.copy()
                out[p + "ifts_head"], out[p + "zfts_head"] = iF[:, :16].copy(), zF[:, :16].copy()
                out[p + "ifts_rowsum"], out[p + "zfts_rowsum"] = iF.astype(np.float64).sum(1), zF.astype(np.float64).sum(1)
                if t == steps - 1:
                    out[p + "ifts"], out[p + "zfts"] = iF.copy(), zF.copy()
                if er is not None:
                    for k_, short in [("batch_instance_relative_position", "env_irel"), ("batch_zone_relative_position", "env_zrel")]:
                        out[p + short] = er[k_][b].numpy().copy()
                    out[p + "env_ifts_head"] = er["batch_instance_fts"][b].numpy()[:, :16].copy()
                    out[p + "env_zfts_head"] = er["batch_zone_fts"][b].numpy()[:, :16].copy()
            print(name, "step", t, [len(F.global_instance_to_patch_dict[b]) for b in range(B)], flush=True)
        np.savez_compressed(os.path.join(OUT, f"g4_{name}.npz"), **out)
    print("g4 ok")


def synth_masks(rng, n, H, W):
    """Overlapping random rectangles / discs as FastSAM 'everything' masks: float32 {0,1}, (n,H,W)."""
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((n, H, W), np.float32)
    for i in range(n):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        if i % 2:
            out[i] = ((yy - cy) ** 2 + (xx - cx) ** 2 < rng.uniform(0.05, 0.4) ** 2 * H * W).astype(np.float32)
        else:
            h, w = rng.uniform(0.1, 0.6) * H, rng.uniform(0.1, 0.6) * W
            out[i] = ((np.abs(yy - cy) < h / 2) & (np.abs(xx - cx) < w / 2)).astype(np.float32)
    return out


def g10_patch_segm():
    """a6: the reference's own `get_patch_segm` (VLN-FF:399-430) with FastSAM's output injected: 'last mask wins' label image ->
    nearest 24x24 -> dense relabel in torch.unique order; a failing segmenter -> zeros."""
    m = rh.load_ref_module("vln")
    sys.argv = ["x"]
    F = m.Feature_Fields(batch_size=1, device="cpu")
    rng = np.random.default_rng(1000)
    cases = {}
    queue = []

    class Prompt:
        def __init__(self, *a, **k):
            pass

        def everything_prompt(self):
            mk = queue.pop(0)
            if mk is None:
                raise RuntimeError("no masks")
            return torch.from_numpy(mk)

    m.FastSAMPrompt = Prompt
    F.FastSAM = lambda *a, **k: None
    specs = [(5, 224, 224), (40, 224, 224), (1, 100, 60), (17, 480, 640), (0, 64, 64), (90, 336, 336), (3, 24, 24), (12, 37, 51)]
    for i, (n, H, W) in enumerate(specs):
        mk = synth_masks(rng, n, H, W) if n else None
        if i == 2:
            mk[:] = 0                                                    # one all-empty mask
        queue.append(mk)
        out = m.Feature_Fields.get_patch_segm(F, [np.zeros((H, W, 3), np.uint8)])
        cases[f"masks_{i}"] = np.zeros((0, H, W), np.uint8) if mk is None else mk.astype(np.uint8)
        cases[f"segm_{i}"] = out.numpy()
    cases["n"] = np.int64(len(specs))
    np.savez_compressed(os.path.join(OUT, "g10_patch_segm.npz"), **cases)
    print("g10 ok", [int(cases[f"segm_{i}"].max()) + 1 for i in range(len(specs))])


def g10b_patch_segm_batch():
    """a6 at the step's own size (round 6): ONE call of the reference's `get_patch_segm` (VLN-FF:399-430) on B * V = 8 images
    of 336 x 336 with FastSAM-shaped outputs injected -- 40 to 120 overlapping masks per image, one image whose segmenter call fails (the
    `except` branch, VLN-FF:424-426: an all-zero map).  The masks are stored bit-packed (inputs are data); the expected label maps are the
    reference function's return value."""
    m = rh.load_ref_module("vln")
    sys.argv = ["x"]
    F = m.Feature_Fields(batch_size=1, device="cpu")
    rng = np.random.default_rng(1010)
    queue = []

    class Prompt:
        def __init__(self, *a, **k):
            pass

        def everything_prompt(self):
            mk = queue.pop(0)
            if mk is None:
                raise RuntimeError("no masks")
            return torch.from_numpy(mk)

    m.FastSAMPrompt = Prompt
    F.FastSAM = lambda *a, **k: None
    counts = [40, 64, 90, 0, 41, 120, 55, 48]
    H = W = 336
    stacks = [synth_masks(rng, n, H, W) if n else None for n in counts]
    queue.extend(stacks)
    out = m.Feature_Fields.get_patch_segm(F, [np.zeros((H, W, 3), np.uint8)] * len(counts))          # (8, 1, 24, 24) int64
    allm = np.concatenate([s.astype(np.uint8) for s in stacks if s is not None])
    np.savez_compressed(os.path.join(OUT, "g10b_patch_segm_batch.npz"), counts=np.asarray(counts, np.int64), H=np.int64(H), W=np.int64(W),
                        masks_packed=np.packbits(allm.reshape(-1)), segm=out.numpy())
    print("g10b ok", [int(out[i].max()) + 1 for i in range(len(counts))])


def g7_text_to_action():
    """Executes the reference's own `convert_text_to_action` (VLN-POL:472-506): the module cannot
    be imported (habitat/gym/cv2/peft), so the single FunctionDef is located with `ast` and
    compiled in place from the mounted file -- nothing is copied into the repo."""
    import ast
    import json
    path = os.path.join(rh.REF_ROOT, "Dynam3D_VLN/vlnce_baselines/models/Policy_Dynam3D_VLN.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "convert_text_to_action"][0]
    ns = {"math": math}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    f = ns["convert_text_to_action"]
    texts = ["turn left 3 steps, move 5 steps.", "turn right 2 steps, move 1 steps.", "turn left 4 steps, move 5 steps.",
             "turn left 9 steps, move 5 steps.", "turn right 0 steps, move 12 steps.", "stop.", "error.",
             "turn left 3 steps", "turn right 3", "turn left 1 steps, move 0 steps.", "turn right 4 steps, move 3 steps.",
             "turn left 2 steps, stop.", "turn right 3 steps, move 2 steps", "turn left 0 steps, move 8 steps.<|end|>"]
    rows = []
    for t in texts:
        try:
            r = f(None, [t])[0]
        except Exception as e:  # the reference raises on some malformed strings; recorded as such
            r = "raises:" + type(e).__name__
        rows.append([t, r if not isinstance(r, tuple) else list(r)])
    with open(os.path.join(OUT, "g7_text_to_action.json"), "w") as fo:
        json.dump(rows, fo, indent=1)
    print("g7 ok", rows)


def g20_gt_text():
    """Executes the reference's own `get_gt_text` (VLN-POL:294-327) -- located with `ast` and compiled in place, like g7 -- on seeded
    (angle, distance, stop) batches with histories that exercise the capped-turn carry-over (`keep_target_waypoint`) and the
    repeated-history "error" rule.  Data only: inputs, returned sentences, the carried waypoints."""
    import ast
    import json
    from types import SimpleNamespace
    path = os.path.join(rh.REF_ROOT, "Dynam3D_VLN/vlnce_baselines/models/Policy_Dynam3D_VLN.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "get_gt_text"][0]
    ns = {"math": math, "np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    f = ns["get_gt_text"]
    rng = np.random.default_rng(200)
    hist_pool = ["none\n", "turn left 4 steps, move 3 steps.\n", "turn right 2 steps, move 1 steps.\n", "turn left 1 steps, move 2 steps.\n",
                 "turn right 4 steps, move 0 steps.\n", "stop.\n", "turn left 0 steps, move 5 steps.\n"]
    cases = []
    for c in range(60):
        B = int(rng.integers(1, 5))
        if c < 30:
            angles = [float(rng.uniform(0, 2 * math.pi)) for _ in range(B)]
        else:                                   # exact multiples / half steps of 15 degrees: the banker's-rounding corners
            angles = [float(np.radians(rng.choice([0, 7.5, 15, 22.5, 52.5, 60, 67.5, 172.5, 180, 187.5, 300, 352.5, 359.6]))) for _ in range(B)]
        dists = [float(rng.choice([0.25, 0.5, 0.75, 1.0, 1.1, 2.25, 2.9, 0.125, 0.375])) for _ in range(B)]
        stops = [bool(rng.random() < 0.15) for _ in range(B)]
        if c % 3 == 0:                          # histories that repeat one sentence -> the "error" rule can fire
            rep = str(rng.choice(hist_pool[1:5]))
            hist = [[rep, rep, rep, rep] for _ in range(B)]
            if c % 2 == 0:                      # the angle whose sentence starts like the repeated history entry (only the LAST row can fire)
                angles = [float(np.radians({1: 60.0, 2: 330.0, 3: 15.0, 4: 300.0}[hist_pool.index(rep)]))] * B
                stops = [False] * B
        else:
            hist = [[str(rng.choice(hist_pool)) for _ in range(4)] for _ in range(B)]
        me = SimpleNamespace(feature_fields=SimpleNamespace(keep_target_waypoint=[None] * B, history_actions=[list(h) for h in hist]))
        out = f(me, list(angles), list(dists), list(stops))
        keep = [None if k is None else [float(k[0]), float(k[1])] for k in me.feature_fields.keep_target_waypoint]
        cases.append(dict(angles=angles, distances=dists, stops=stops, history=hist, text=out, keep=keep))
    with open(os.path.join(OUT, "g20_gt_text.json"), "w") as fo:
        json.dump(cases, fo, indent=0)
    print("g20 ok", sum(t == "error.<|end|>" for c in cases for t in c["text"]), "error sentences,",
          sum(k is not None for c in cases for k in c["keep"]), "carried waypoints")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g7"]
    torch.set_num_threads(8)
    fns = {"g1": g1_unproject, "g2": g2_frustum, "g2b": g2b_frustum_pinhole, "g3": g3_knn, "g4": g4_trajectories, "g7": g7_text_to_action, "g10": g10_patch_segm, "g10b": g10b_patch_segm_batch, "g20": g20_gt_text}
    for w in which:
        fns[w]()
