"""CPU: intrinsics / extrinsics path -- oracle vs the reference-generated golden, host logic vs oracle."""
import numpy as np

from oracle import geometry as G
from tests.cpu_ops import CpuOps
from tests.golden_io import load
from tests.pinhole_parity import run_pinhole_vs_oracle


def test_oracle_frustum_mask_pinhole_matches_reference_golden():
    g = load("g2b_frustum_pinhole.npz")
    for i in range(int(g["n"])):
        m = G.frustum_mask_pinhole(g[f"pts_{i}"], g[f"depth_{i}"], g[f"K_{i}"], g[f"view_{i}"], 0.0, 3.0, 0.1)
        assert np.array_equal(m, g[f"mask_{i}"]) and m.sum() > 20
    assert np.array_equal(G.heading_angle(g["heading_pts"].copy()), g["heading_out"])


def test_project_depth_to_3d_restatement_properties():
    """Open3D is absent (parity unpinned): check the restated algorithm against its definition -- re-projecting the points with the
    intrinsics gives back the sampled pixel centres, z is the metric depth; one invalid pixel -> all zeros (PRE-FF:90-91)."""
    rng = np.random.default_rng(0)
    K = np.array([[70.0, 0, 40.5], [0, 72.0, 29.5], [0, 0, 1]])
    d = rng.integers(500, 4000, (60, 80)).astype(np.float32)
    pts, mask = G.project_depth_to_3d(d, K, 1000.0, 1000.0)
    ri, ci = G.torch_nearest_indices(60, 24), G.torch_nearest_indices(80, 24)
    assert np.allclose(pts[:, 2].reshape(24, 24), d[np.ix_(ri, ci)] / 1000.0, rtol=1e-7) and mask.all()
    u = pts[:, 0] / pts[:, 2] * K[0, 0] + K[0, 2]
    v = pts[:, 1] / pts[:, 2] * K[1, 1] + K[1, 2]
    assert np.allclose(u.reshape(24, 24), ci[None, :], atol=1e-9) and np.allclose(v.reshape(24, 24), ri[:, None], atol=1e-9)
    d2 = d.copy()
    d2[3, 4] = 0                                             # zero -> image max: still valid
    assert G.project_depth_to_3d(d2, K)[0][:, 2].min() > 0
    assert not G.project_depth_to_3d(np.zeros((60, 80), np.float32), K)[0].any()                 # all invalid -> zeros
    assert not G.project_depth_to_3d(d, K, 1000.0, 2.0)[0].any()                                   # one truncated pixel -> zeros


def test_pinhole_host_logic_matches_oracle():
    run_pinhole_vs_oracle(CpuOps(), "cpu")
