"""CPU: the product's host orchestration + C++ bookkeeping (csrc/ff_state.cpp) replayed against the
reference-generated golden trajectories, with the HIP kernels swapped for tests/cpu_ops.py."""
import pytest

from tests.cpu_ops import CpuOps
from tests.ff_parity import run_case
from tests.golden_io import TRAJ_CASES


@pytest.mark.parametrize("name", list(TRAJ_CASES))
def test_host_state_machine_matches_golden(name):
    run_case(name, CpuOps(), "cpu")
