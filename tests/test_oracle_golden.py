"""CPU: the oracle (oracle/) against the golden vectors produced by the reference's own code."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle.ff_oracle import FeatureFieldsOracle
from tests.golden_io import TRAJ_CASES, GOLDEN_DIR, int_hash, load, pack_ragged, traj_inputs, unpack_ragged
from dynam3d_amd.weights import ff_param_spec, synth_state_dict


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_g1_unproject_bit_exact():
    g = load("g1_unproject.npz")
    for i in range(int(g["n"])):
        pos, d, s = G.unproject_habitat(g[f"depth_{i}"], g[f"position_{i}"].tolist(), float(g[f"heading_{i}"]))
        assert np.array_equal(bits(pos), bits(g[f"pos_{i}"]))
        assert np.array_equal(bits(d), bits(g[f"dir_{i}"]))
        assert np.array_equal(bits(s), bits(g[f"scale_{i}"]))
    out = G.patch_3d_info(g["info_depth"])
    for n, o in zip(["rel_x", "rel_y", "rel_z", "direction", "scale"], out):
        assert np.array_equal(bits(o), bits(g["info_" + n])), n


def test_g2_frustum_bit_exact():
    g = load("g2_frustum.npz")
    for i in range(int(g["n"])):
        m = G.frustum_mask_habitat(g[f"pts_{i}"], g[f"depth_{i}"], g[f"position_{i}"].tolist(), float(g[f"heading_{i}"]))
        assert np.array_equal(m, g[f"mask_{i}"])
        assert m.sum() > 0


def test_g3_knn_bit_exact():
    g = load("g3_knn.npz")
    for i in range(int(g["n"])):
        d2, idx = G.knn_bruteforce(g[f"pts_{i}"], g[f"q_{i}"], int(g[f"k_{i}"]))
        assert np.array_equal(idx, g[f"idx_{i}"])
        assert np.array_equal(bits(d2), bits(g[f"d2_{i}"]))


def check_env_against_golden(g, t, b, owner, members, zmembers, zkey, ipos, ifts, zpos, zfts, rows_pos, env, last):
    """Shared by the oracle test here and the GPU parity test: exact bookkeeping, toleranced floats."""
    p = f"t{t}_b{b}_"
    ks = np.array(sorted(owner.keys()), np.int64)
    owner_inst = np.array([owner[k] for k in ks.tolist()], np.int64)
    assert np.array_equal(np.array(list(members.keys()), np.int64), g[p + "inst_order"]), (t, b)
    assert np.array_equal(np.array(list(zmembers.keys()), np.int64), g[p + "zone_order"]), (t, b)
    if p + "book_hash" in g.files:      # light step of a long trajectory: digests of the same tables (tests/golden/gen_golden.py)
        im, imo = pack_ragged([np.asarray(m, np.int64) for m in members.values()])
        zmm, zmo = pack_ragged([np.asarray(m, np.int64) for m in zmembers.values()])
        assert int(g[p + "n_rows"]) == rows_pos.shape[0], (t, b)
        assert int_hash(ks, owner_inst, im, imo, zmm, zmo) == int(g[p + "book_hash"]), (t, b)
        assert int_hash(np.ascontiguousarray(rows_pos, np.float32).view(np.uint32)) == int(g[p + "rows_hash"]), (t, b)
    else:
        assert np.array_equal(ks, g[p + "owner_ids"])
        assert np.array_equal(owner_inst, g[p + "owner_inst"])
        for a, e in zip(unpack_ragged(g[p + "inst_members"], g[p + "inst_members_off"]), members.values()):
            assert np.array_equal(a, np.asarray(e, np.int64))
        for a, e in zip(unpack_ragged(g[p + "zone_members"], g[p + "zone_members_off"]), zmembers.values()):
            assert np.array_equal(a, np.asarray(e, np.int64))
        assert np.array_equal(bits(rows_pos), bits(g[p + "rows_pos"]))
    assert np.array_equal(np.array(list(zkey.keys()), np.float32).reshape(-1, 3), g[p + "zone_keys"])
    assert np.array_equal(np.array(list(zkey.values()), np.int64), g[p + "zone_key_ids"])
    # merged centroids average tomb-stoned rows (-1e4) in float32 in the reference (F11): tolerance
    # is relative to the magnitude of the summands.
    tol = lambda ref: 2e-6 * np.maximum(1.0, np.abs(ref)) + 2e-7 * 1e4
    assert ipos.shape == g[p + "ipos"].shape
    assert np.all(np.abs(ipos - g[p + "ipos"]) <= tol(g[p + "ipos"]))
    assert zpos.shape == g[p + "zpos"].shape
    gz = g[p + "zpos"]
    assert np.array_equal(np.isnan(zpos), np.isnan(gz))
    assert np.all(np.abs(np.nan_to_num(zpos) - np.nan_to_num(gz)) <= tol(np.nan_to_num(gz)))
    assert np.allclose(ifts[:, :16], g[p + "ifts_head"], atol=1e-3, rtol=1e-3)
    assert np.allclose(zfts[:, :16], g[p + "zfts_head"], atol=1e-3, rtol=1e-3)
    assert np.allclose(ifts.astype(np.float64).sum(1), g[p + "ifts_rowsum"], atol=2e-2)
    if last:
        rel = np.linalg.norm(ifts - g[p + "ifts"]) / max(np.linalg.norm(g[p + "ifts"]), 1e-9)
        assert rel < 1e-3, rel
        rel = np.linalg.norm(zfts - g[p + "zfts"]) / max(np.linalg.norm(g[p + "zfts"]), 1e-9)
        assert rel < 1e-3, rel
    if p + "env_irel" not in g.files:                # Pretrain-class goldens: PRE-FF has no get_environment_features
        return
    assert env["irel"].shape == g[p + "env_irel"].shape
    assert np.all(np.abs(env["irel"] - g[p + "env_irel"]) <= tol(g[p + "env_irel"]))
    assert env["zrel"].shape == g[p + "env_zrel"].shape
    assert np.allclose(env["ifts"][:, :16], g[p + "env_ifts_head"], atol=1e-3, rtol=1e-3)
    assert np.allclose(env["zfts"][:, :16], g[p + "env_zfts_head"], atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("name", list(TRAJ_CASES))
def test_g4_trajectory(name):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    case = TRAJ_CASES[name]
    g = load(f"g4_{name}.npz")
    sd = synth_state_dict(ff_param_spec(), seed=0)
    orc = FeatureFieldsOracle(sd, case["B"], num_proposals=4 if case.get("variant") == "pretrain" else 2)
    V, vid = case.get("views", 1), case.get("view_ids")
    for t, inp in enumerate(traj_inputs(case)):
        if case.get("pop") and case["pop"][0] == t:
            orc.pop(case["pop"][1])
        orc.delete_old_features_from_camera_frustum(inp["depth_full"], inp["positions"], inp["headings"], num_of_views=V, view_ids=vid)
        orc.update_feature_fields(inp["depth24"], inp["grid"], inp["patch_segm"], inp["positions"], inp["headings"], num_of_views=V,
                                  view_ids=vid)
        ev = orc.get_environment_features(inp["positions"], inp["headings"])
        assert len(orc.env) == len(inp["alive"])
        for b, e in enumerate(orc.env):
            env = dict(irel=ev["batch_instance_relative_position"][b], zrel=ev["batch_zone_relative_position"][b],
                       ifts=ev["batch_instance_fts"][b], zfts=ev["batch_zone_fts"][b])
            check_env_against_golden(g, t, b, e.owner, e.members, e.zmembers, e.zkey, e.ipos, e.ifts, e.zpos, e.zfts,
                                     e.pos, env, t == case["steps"] - 1)


def test_preprocess_depth_and_resize():
    rng = np.random.default_rng(0)
    d = rng.uniform(0.05, 0.5, (3, 37, 41, 1)).astype(np.float32)
    d[rng.random(d.shape) < 0.05] = 0
    t = torch.from_numpy(d) * 1.0
    mx, _ = t.max(dim=1, keepdim=True)
    mx = mx.expand(-1, 37, -1, -1)
    t[t == 0] = mx[t == 0]
    t = (0.0 * 100.0 + t * (10.0 - 0.0) * 100.0) / 100.0          # VLN-POL:184-185 arithmetic
    assert np.array_equal(bits(G.preprocess_depth(d)), bits(t.numpy()))
    idx = G.nearest_indices(256, 24)
    assert idx[0] == 0 and idx[-1] == int(math.floor(23 * 256 / 24)) and len(idx) == 24
