"""GPU: the device-resident planner of the memory update (csrc/ff_plan_kernels.hip, one workgroup per environment) -- against the
reference-generated golden trajectories, against the host state machine on random decision streams (tests/ff_plan_diff.py), and the
whole feature field bit for bit against its host-planned twin."""
import numpy as np
import pytest
import torch

from tests.ff_parity import run_case
from tests.golden_io import TRAJ_CASES, traj_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(TRAJ_CASES))
def test_device_planner_trajectory_parity(name):
    from dynam3d_amd.ops import HipOps
    ff = run_case(name, HipOps(), "cuda", planner="device")
    assert ff.state.hdr.is_cuda


def test_device_planner_pools_grow():
    from dynam3d_amd.ops import HipOps
    ff = run_case("walk", HipOps(), "cuda", max_steps=1, m_cap=8, z_cap=2, planner="device")
    assert ff.pools.n_cap >= 7 * 576 and ff.pools.m_cap > 8 and ff.pools.z_cap > 2 and ff.state.R == ff.pools.n_cap


@pytest.mark.parametrize("compat", ["reference", "fixed"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_planner_kernels_equal_host_state_machine_on_random_streams(compat, seed):
    from dynam3d_amd.ops import HipOps
    from tests.ff_plan_diff import run_random
    ops = HipOps()
    wide = seed % 2 == 1
    stats = run_random(ops, ops.lib, "cuda", compat, seed, K=4 if wide else 2, k_max=4 if wide else 2, P=576 if seed == 3 else 48, steps=12 if seed == 3 else 25,
                       max_seg=40 if seed == 3 else 9)
    assert stats["dead_inst"] > 0 and stats["merges"] > 0


def test_device_and_host_planned_fields_are_bit_identical():
    """Two feature fields over the same frames, one planned on the host, one on the device: every store, every feature, every output
    of get_environment_features is the same bit pattern (the float kernels see the same index tables in the same order)."""
    from dynam3d_amd.feature_fields import Feature_Fields
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    case = TRAJ_CASES["walk"]
    sd = synth_state_dict(ff_param_spec(), seed=0)
    a = Feature_Fields(case["B"], "cuda", sd, planner="host")
    b = Feature_Fields(case["B"], "cuda", sd, planner="device")
    for f in (a, b):
        f.initialize_camera_setting(90.0, 90.0)
    for t, inp in enumerate(traj_inputs(case)):
        outs = []
        for f in (a, b):
            f.delete_old_features_from_camera_frustum(torch.from_numpy(inp["depth_full"]), inp["positions"], inp["headings"])
            f.update_feature_fields(inp["depth24"], inp["grid"], None, inp["positions"], inp["headings"], patch_segm=inp["patch_segm"])
            outs.append(f.get_environment_features(inp["positions"], inp["headings"]))
        for key in outs[0]:
            for x, y in zip(outs[0][key], outs[1][key]):
                assert torch.equal(x, y), (t, key)
        for e in range(case["B"]):
            ea, eb = a.export_env(e), b.export_env(e)
            assert ea["owner"] == eb["owner"] and list(ea["members"]) == list(eb["members"]) and list(ea["zmembers"]) == list(eb["zmembers"])
            for k in ("ipos", "ifts", "zpos", "zfts", "rows_pos"):
                assert np.array_equal(ea[k], eb[k], equal_nan=True), (t, e, k)


def test_long_episode_twin_with_growing_pools():
    """30 steps, 4 environments walking through walls (deletions every step), instance / zone / edge capacities that start tiny and have
    to double several times: the device-planned field stays bit-identical to the host-planned one at every step."""
    from dynam3d_amd.feature_fields import Feature_Fields
    from dynam3d_amd.synthetic import SyntheticEpisodes
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    B, steps = 4, 30
    sd = synth_state_dict(ff_param_spec(), seed=0)
    a = Feature_Fields(B, "cuda", sd, planner="host", max_steps=4, m_cap=16, z_cap=8)
    b = Feature_Fields(B, "cuda", sd, planner="device", max_steps=4, m_cap=16, z_cap=8)
    b.state.E, b.state.edges, b.state._struct = 64, torch.zeros((B, 2, 2, 64), dtype=torch.int32, device="cuda"), None    # a tiny edge table too
    for f in (a, b):
        f.initialize_camera_setting(90.0, 90.0)
    ep = SyntheticEpisodes(B, seed=11, wall=2.5)
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(steps):
        fr = ep.next()
        depth = torch.from_numpy(fr.depth).cuda()[..., 0]
        grid = torch.randn(B, 1, 576, 768, device="cuda", generator=g).half()
        pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
        outs = []
        for f in (a, b):
            dfull = f.ops.preprocess_depth(depth).view(B, 1, depth.shape[1], depth.shape[2])
            d24 = f.ops.resize_nearest_preprocess(depth, 24, 24).view(B, 1, 576)
            f.delete_old_features_from_camera_frustum(dfull, pos, hd)
            f.update_feature_fields(d24, grid, None, pos, hd, patch_segm=fr.patch_segm)
            outs.append(f.get_environment_features(pos, hd))
        for key in outs[0]:
            for x, y in zip(outs[0][key], outs[1][key]):
                assert torch.equal(x, y), (t, key)
        for name in ("inst_pos", "inst_fts", "zone_pos", "zone_fts", "rows_pos"):
            x, y = getattr(a.pools, name), getattr(b.pools, name)
            n = min(x.shape[1], y.shape[1])
            assert torch.equal(x[:, :n].nan_to_num(7.0), y[:, :n].nan_to_num(7.0)), (t, name)
        for e in range(B):
            assert a.state.count(e, a.state.LIVE) == b.state.count(e, b.state.LIVE) and a.state.count(e, a.state.ZLIVE) == b.state.count(e, b.state.ZLIVE)
    assert b.pools.m_cap > 16 and b.pools.n_cap > 4 * 576 and b.state.M == b.pools.m_cap and b.state.E > 64
    assert max(a.state.count(e, a.state.OWNED) for e in range(B)) < steps * 576          # the walls did delete patches
