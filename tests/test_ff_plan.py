"""CPU: the device planner's source (csrc/ff_plan.h, compiled over host arrays into the test library) against the reference-generated
golden trajectories and, on random decision streams, against the host state machine (csrc/ff_state.cpp)."""
import pytest

from tests.cpu_ops import CpuOps
from tests.ff_parity import run_case
from tests.ff_plan_diff import run_random
from tests.golden_io import TRAJ_CASES


@pytest.mark.parametrize("name", list(TRAJ_CASES))
def test_device_planner_matches_golden(name):
    run_case(name, CpuOps(), "cpu", planner="device")


def test_device_planner_pools_grow():
    ff = run_case("walk", CpuOps(), "cpu", max_steps=1, planner="device")
    assert ff.pools.n_cap >= 7 * 576 and ff.state.R == ff.pools.n_cap
    ff = run_case("walk", CpuOps(), "cpu", m_cap=8, z_cap=2, planner="device")
    assert ff.pools.m_cap > 8 and ff.pools.z_cap > 2 and ff.state.M == ff.pools.m_cap and ff.state.Z == ff.pools.z_cap


@pytest.mark.parametrize("compat", ["reference", "fixed"])
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_planner_equals_host_state_machine_on_random_streams(compat, seed):
    ops = CpuOps()
    stats = run_random(ops, ops.lib, "cpu", compat, seed, K=2 if seed % 2 == 0 else 4, k_max=2 if seed % 2 == 0 else 4)
    assert stats["dead_inst"] > 0 and stats["merges"] > 0                        # the stream exercised deletions and merges


def test_random_streams_cover_the_corner_cases():
    ops = CpuOps()
    tot = {}
    for seed in range(4):
        for k, v in run_random(ops, ops.lib, "cpu", "reference", 10 + seed, steps=30).items():
            tot[k] = tot.get(k, 0) + v
    assert all(tot[k] > 0 for k in ("dead_inst", "dead_zone", "merges", "multi", "shrink", "recycled")), tot


def test_planner_reports_a_dead_proposal_like_the_host_raises():
    """A merge decision that points at a slot which is not a live instance is a KeyError in the reference and an error code in the host
    state machine; the device planner raises it through the view's report."""
    import numpy as np
    import torch

    from dynam3d_amd.ff_plan import REPORT_WORDS, DevicePlanner
    ops, P = CpuOps(), 8
    dev = DevicePlanner("reference", P, 2, "cpu")
    dev.reset(1, 4 * P, 16, 16, (-5000, -5000, -5000))
    slot = torch.zeros(1, dtype=torch.int32)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32))
    order, seg = i32(np.arange(P)[None]), i32(np.zeros((1, P)))
    off, nseg, cells = i32([[0, P]]), i32([1]), i32([[0, 0, 0]])

    def view(idx_val, logit):
        k0, _ = ops.ffdev_begin_view(dev, slot)
        rep = torch.zeros(REPORT_WORDS, dtype=torch.int32)
        d2 = torch.zeros((1, 1, 2)); idx = i32(np.full((1, 1, 2), idx_val)); lg = torch.tensor([[[[0.0, logit], [0.0, -1.0]]]])
        out = ops.ffdev_plan_merge(dev, slot, order, seg, off, nseg, 1, 2, k0, d2, idx, lg, cells, 4 * P, rep)
        mc = torch.zeros((1, 3), dtype=torch.int32)
        ops.ffdev_plan_zones(dev, slot, out[1], mc, cells, nseg, 1, 16, rep)
        dev.n_rows[0] += P
        return rep.numpy().reshape(1, REPORT_WORDS)

    dev.take_report_envs([0], view(0, -1.0))                 # first view: no tree yet, the segment opens instance 0
    dev.take_report_envs([0], view(0, 1.0))                  # merges into instance 0: fine
    assert dev.n_live[0] == 1 and dev.n_owned[0] == 2 * P
    with pytest.raises(RuntimeError, match="not a live instance"):
        dev.take_report_envs([0], view(5, 1.0))              # slot 5 was never opened
