"""CPU: the product's host orchestration + C++ bookkeeping (csrc/ff_state.cpp) replayed against the
reference-generated golden trajectories, with the HIP kernels swapped for tests/cpu_ops.py."""
import pytest

from tests.cpu_ops import CpuOps
from tests.ff_parity import run_case
from tests.golden_io import TRAJ_CASES


@pytest.mark.parametrize("name", list(TRAJ_CASES))
def test_host_state_machine_matches_golden(name):
    run_case(name, CpuOps(), "cpu", planner="host")


def test_row_pools_grow_mid_episode():
    """Capacity for ONE step of rows: every later step reallocates the row pools (Feature_Fields._grow_rows) and must carry the
    stored rows, tomb-stones and features over unchanged -- the whole 'walk' trajectory still matches the reference golden."""
    ff = run_case("walk", CpuOps(), "cpu", max_steps=1, planner="host")
    assert ff.pools.n_cap >= 7 * 576


def test_instance_and_zone_pools_grow():
    """The reference's instance / zone stores are unbounded (torch.cat); tiny initial pools must double their way through the golden."""
    ff = run_case("walk", CpuOps(), "cpu", m_cap=8, z_cap=2, planner="host")
    assert ff.pools.m_cap > 8 and ff.pools.z_cap > 2
