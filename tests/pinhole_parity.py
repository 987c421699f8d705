"""Shared driver: product Feature_Fields in the intrinsics / extrinsics mode (any ops backend) vs the oracle."""
import numpy as np
import torch

from dynam3d_amd.feature_fields import Feature_Fields
from dynam3d_amd.weights import ff_param_spec, synth_state_dict
from oracle.ff_oracle import FeatureFieldsOracle
from tests.pinhole_scene import frames


def run_pinhole_vs_oracle(ops, device, B=2, V=2, steps=3):
    sd = synth_state_dict(ff_param_spec(), seed=0)
    ff = Feature_Fields(B, device=device, state_dict=sd, ops=ops, max_steps=steps + 1, max_views=V, variant="pretrain")
    orc = FeatureFieldsOracle(sd, B, num_proposals=4)
    rng = np.random.default_rng(3)
    culled = 0
    for t, fr in enumerate(frames(B, V, steps)):
        grid = rng.standard_normal((B, V, 576, 768)).astype(np.float32)
        before = [int((e.pos[:, 0] > -9000).sum()) for e in orc.env]
        ff.delete_old_features_from_camera_frustum(torch.from_numpy(fr["depth_m"]), batch_camera_intrinsic=torch.from_numpy(fr["intrinsics"]),
                                                   batch_extrinsic=torch.from_numpy(fr["extrinsic"]))
        orc.delete_old_features_from_camera_frustum(fr["depth_m"], batch_camera_intrinsic=fr["intrinsics"], batch_extrinsic=fr["extrinsic"])
        culled += sum(before) - sum(int((e.pos[:, 0] > -9000).sum()) for e in orc.env)
        ff.update_feature_fields([d for d in fr["depth_raw"]], grid, None, batch_camera_intrinsic=fr["intrinsics"], batch_rot=fr["rot"],
                                 batch_trans=fr["trans"], patch_segm=fr["patch_segm"])
        orc.update_feature_fields(fr["depth_raw"], grid, fr["patch_segm"], batch_camera_intrinsic=fr["intrinsics"], batch_rot=fr["rot"],
                                  batch_trans=fr["trans"])
        for b in range(B):
            ex, e = ff.export_env(b), orc.env[b]
            # world positions / directions pass through float64 (R @ p + T, arcsin) rounded once: 1 ulp of float32
            assert np.allclose(ex["rows_pos"], e.pos, rtol=2e-7, atol=1e-6), np.abs(ex["rows_pos"] - e.pos).max()
            assert ex["owner"] == e.owner and list(ex["members"]) == list(e.members)
            assert all(np.array_equal(ex["members"][k], e.members[k]) for k in e.members)
            assert list(ex["zkey"].items()) == list(e.zkey.items())
            assert np.allclose(ex["ipos"], e.ipos, atol=2e-3) and np.allclose(ex["ifts"], e.ifts, atol=5e-3)
    assert culled > 0, "the scene must exercise the cull"
    return ff
