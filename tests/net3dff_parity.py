"""Shared driver: product `Net_3DFF` front-end (any ops backend/device) vs `oracle.step_oracle.Net3DFFOracle` on a synthetic
12-sensor panorama walk.  Bookkeeping must agree exactly; floats within the tolerances of the policy tests."""
import numpy as np
import torch

from dynam3d_amd.net_3dff import Net_3DFF
from dynam3d_amd.policy import synth_policy_weights
from dynam3d_amd.synthetic import SyntheticEpisodes
from oracle.step_oracle import Net3DFFOracle


def pano_observations(eps):
    """12 sensors in the counter-clockwise key order Habitat hands over: 'rgb','depth','rgb_1','depth_1',... (key a = view a)."""
    frs = [ep.next() for ep in eps]
    obs = {}
    for a, fr in enumerate(frs):
        sfx = "" if a == 0 else f"_{a}"
        obs["rgb" + sfx], obs["depth" + sfx] = fr.rgb, fr.depth
    return obs, frs


def run_net3dff_vs_oracle(ops, device, cfg, steps=2, B=2, clip_dtype=torch.float32, fts_tol=2e-3):
    import dataclasses
    from tests.test_policy_cpu import toy_dense
    with toy_dense(dataclasses.replace(cfg, clip_dtype=clip_dtype), device):
        return _run_net3dff_vs_oracle(ops, device, cfg, steps, B, clip_dtype, fts_tol)


def _run_net3dff_vs_oracle(ops, device, cfg, steps, B, clip_dtype, fts_tol):
    sd = synth_policy_weights(cfg, seed=0)
    net = Net_3DFF(cfg.vit, sd, device=device, batch_size=B, ops=ops, clip_dtype=clip_dtype, max_steps=steps + 1)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    orc = Net3DFFOracle(sd, cfg.vit, B)
    eps = [SyntheticEpisodes(B, seed=30 + a, image_hw=112, depth_hw=64) for a in range(12)]
    for t in range(steps):
        obs, frs = pano_observations(eps)
        pos, hd = [p.tolist() for p in frs[0].positions], list(frs[0].headings)
        # clockwise slot v <- key (12 - v) % 12; segmentation of the kept views, environment-major
        segm = np.stack([frs[(12 - v) % 12].patch_segm for v in (0, 3, 6, 9)], 1).reshape(B * 4, 1, 24, 24)
        net.positions, net.headings = pos, hd
        out = net({k: torch.from_numpy(v) for k, v in obs.items()}, patch_segm=segm)
        ref = orc.forward(obs, pos, hd, segm)
        assert np.array_equal(out["depth24"].cpu().numpy(), ref["depth24"])
        g, r = out["grid_fts"].float().cpu().numpy(), ref["grid_fts"]
        assert np.linalg.norm(g - r) / np.linalg.norm(r) < fts_tol
        for b in range(B):
            ex, e = net.feature_fields.export_env(b), orc.ff.env[b]
            if clip_dtype == torch.float32:                       # same features -> same merge decisions -> same bookkeeping
                assert ex["owner"] == e.owner and list(ex["members"]) == list(e.members)
                assert all(np.array_equal(ex["members"][k], e.members[k]) for k in e.members)
                assert list(ex["zkey"].items()) == list(e.zkey.items())
                assert np.allclose(ex["ipos"], e.ipos, atol=2e-3) and np.allclose(ex["ifts"], e.ifts, atol=5e-3)
            assert np.array_equal(ex["rows_pos"], e.pos)       # unprojection of the four views: bit-exact
    return net
