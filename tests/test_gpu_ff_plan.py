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
