"""CPU: host logic of the Pretrain multi-view front-end (`dynam3d_amd.net_3dff.Net_3DFF`, PRE-POL:136-189) against the oracle,
HIP kernels swapped for tests/cpu_ops.py."""
import numpy as np
import pytest
import torch

from dynam3d_amd.net_3dff import clockwise_sources
from tests.cpu_ops import CpuOps
from tests.net3dff_parity import run_net3dff_vs_oracle
from tests.test_policy_cpu import SMALL


def test_clockwise_sources():
    obs = {}
    for a in range(12):
        sfx = "" if a == 0 else f"_{a}"
        obs["rgb" + sfx] = obs["depth" + sfx] = None
    assert clockwise_sources(obs) == ["depth", "depth_9", "depth_6", "depth_3"]             # slot v <- key (12 - v) % 12
    assert clockwise_sources(obs, range(12))[1] == "depth_11"
    with pytest.raises(ValueError):
        clockwise_sources({"depth": None, "rgb": None})


def test_net3dff_front_end_matches_oracle():
    run_net3dff_vs_oracle(CpuOps(), "cpu", SMALL)
