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
