"""Container-only: drive the REFERENCE Feature_Fields and the oracle side by side on fresh seeds
(beyond the committed goldens).  Skipped when /root/reference is not mounted (GPU box)."""
import numpy as np
import pytest
import torch

from tests.golden_io import traj_inputs
from dynam3d_amd.weights import ff_param_spec, synth_state_dict


@pytest.mark.reference
@pytest.mark.parametrize("case", [dict(B=1, steps=4, seed=11, grid_seed=12, stationary=False, wall=None, depth_hw=96),
                                  dict(B=1, steps=4, seed=13, grid_seed=14, stationary=True, wall=1.5, depth_hw=48),
                                  # 30 live steps, fresh seed, wall steps: deletions, recycled ids and stale zone snapshots at bench-like depth
                                  dict(B=2, steps=30, seed=31, grid_seed=32, stationary=False, wall=2.0, wall_steps=(7, 8, 16, 24), depth_hw=64)])
def test_oracle_matches_reference_live(case):
    from oracle import ref_harness as rh
    from oracle.ff_oracle import FeatureFieldsOracle
    sd = synth_state_dict(ff_param_spec(), seed=0)
    B = case["B"]
    ref = rh.RefFeatureFields(B, sd)
    orc = FeatureFieldsOracle(sd, B)
    for inp in traj_inputs(case):
        ref.step(torch.from_numpy(inp["depth_full"]), inp["depth24"], inp["grid"], torch.from_numpy(inp["patch_segm"]),
                 inp["positions"], inp["headings"])
        orc.delete_old_features_from_camera_frustum(inp["depth_full"], inp["positions"], inp["headings"])
        orc.update_feature_fields(inp["depth24"], inp["grid"], inp["patch_segm"], inp["positions"], inp["headings"])
        F = ref.F
        for b, e in enumerate(orc.env):
            assert dict(F.global_patch_to_instance_dict[b]) == e.owner
            assert list(F.global_instance_to_patch_dict[b].keys()) == list(e.members.keys())
            assert all(np.array_equal(F.global_instance_to_patch_dict[b][k], e.members[k]) for k in e.members)
            assert list(F.global_zone_to_instance_dict[b].keys()) == list(e.zmembers.keys())
            assert dict(F.global_zone_key_to_id[b]) == e.zkey
            assert np.allclose(F.global_instance_fts[b].numpy(), e.ifts, atol=1e-4)
            assert np.allclose(F.global_zone_fts[b].numpy(), e.zfts, atol=1e-4)
