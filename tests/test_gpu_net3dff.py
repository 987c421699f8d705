"""GPU: Pretrain multi-view front-end (PRE-POL:136-189) on the HIP kernels vs the CPU oracle.
float32 tower: bookkeeping exact, features 2e-3; fp16 tower (the reference's dtype): features within 3e-2 (fp16 storage)."""
import dataclasses

import pytest
import torch

from tests.net3dff_parity import run_net3dff_vs_oracle
from tests.test_policy_cpu import SMALL

pytestmark = pytest.mark.gpu


def test_net3dff_fp32_matches_oracle():
    from dynam3d_amd.ops import HipOps
    net = run_net3dff_vs_oracle(HipOps(), "cuda", SMALL, steps=3)
    assert net.feature_fields.state.count(0, net.feature_fields.state.ROWS) == 3 * 4 * 576


def test_net3dff_fp16_tower_close_to_oracle():
    from dynam3d_amd.ops import HipOps
    run_net3dff_vs_oracle(HipOps(), "cuda", SMALL, steps=2, clip_dtype=torch.float16, fts_tol=3e-2)
