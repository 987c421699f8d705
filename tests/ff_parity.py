"""Shared trajectory-parity driver: runs dynam3d_amd.Feature_Fields (any ops backend/device) over a
golden case and checks every step against the reference-generated fixture."""
import numpy as np
import torch

from dynam3d_amd.feature_fields import Feature_Fields
from dynam3d_amd.weights import ff_param_spec, synth_state_dict
from tests.golden_io import TRAJ_CASES, load, traj_inputs
from tests.test_oracle_golden import check_env_against_golden


def run_case(name, ops, device, on_step=None, max_steps=None, m_cap=4096, z_cap=2048, planner=None):
    """max_steps: row-pool capacity in steps (default: the whole trajectory fits); smaller values make the pools GROW mid-episode."""
    from dynam3d_amd.f32_ops import F32Ops
    check_before, F32Ops.CHECK = F32Ops.CHECK, str(device) != "cpu"      # every split-precision float32 GEMM output verified finite (f32_ops.py)
    try:
        return _run_case(name, ops, device, on_step, max_steps, m_cap, z_cap, planner)
    finally:
        F32Ops.CHECK = check_before


def _run_case(name, ops, device, on_step, max_steps, m_cap, z_cap, planner):
    case = TRAJ_CASES[name]
    g = load(f"g4_{name}.npz")
    sd = synth_state_dict(ff_param_spec(), seed=0)
    ff = Feature_Fields(case["B"], device=device, state_dict=sd, ops=ops,
                        max_steps=max_steps or (case["steps"] + 1) * case.get("views", 1), variant=case.get("variant", "vln"),
                        m_cap=m_cap, z_cap=z_cap, planner=planner)
    ff.initialize_camera_setting(90.0, 90.0)
    V, vid = case.get("views", 1), case.get("view_ids")           # view_ids = the Pretrain class's keyword (PRE-FF:674,843)
    for t, inp in enumerate(traj_inputs(case)):
        if case.get("pop") and case["pop"][0] == t:
            ff.pop(case["pop"][1])
        ff.delete_old_features_from_camera_frustum(torch.from_numpy(inp["depth_full"]), inp["positions"], inp["headings"], num_of_views=V,
                                                   view_ids=vid)
        ff.update_feature_fields(inp["depth24"], inp["grid"], None, inp["positions"], inp["headings"], num_of_views=V,
                                 patch_segm=inp["patch_segm"], view_ids=vid)
        ev = ff.get_environment_features(inp["positions"], inp["headings"])
        assert ff.batch_size == len(inp["alive"])
        for b in range(ff.batch_size):
            ex = ff.export_env(b)
            env = dict(irel=ev["batch_instance_relative_position"][b].cpu().numpy(), zrel=ev["batch_zone_relative_position"][b].cpu().numpy(),
                       ifts=ev["batch_instance_fts"][b].cpu().numpy(), zfts=ev["batch_zone_fts"][b].cpu().numpy())
            check_env_against_golden(g, t, b, ex["owner"], ex["members"], ex["zmembers"], ex["zkey"], ex["ipos"], ex["ifts"],
                                     ex["zpos"], ex["zfts"], ex["rows_pos"], env, t == case["steps"] - 1)
        if on_step:
            on_step(t, ff)
    return ff
