"""CPU: render oracle (oracle/render_oracle.py) vs the reference-generated golden g6 (small case; the full 12x12x501 case
is replayed on the GPU by tests/test_gpu_render.py)."""
import numpy as np

from oracle import render_oracle as RO
from tests.golden_io import RENDER_CASES, load, render_scene
from dynam3d_amd.weights import ff_param_spec, render_param_spec, synth_state_dict


def test_render_oracle_matches_reference_small():
    case = RENDER_CASES["small"]
    g = load("g6_render.npz")
    pos, pdir, psc, fts = render_scene(case)
    sd = synth_state_dict(ff_param_spec() + render_param_spec(), seed=0)
    o = RO.render_view(pos, pdir, psc, fts, sd, case["position"], case["heading"], H=case["H"], W=case["W"], n_samples=case["n_samples"])
    ok = g["small_n_ranked"] >= 8
    assert np.array_equal(o["n_ranked"], g["small_n_ranked"]) and ok.any()
    d = np.abs(o["feature_map"].reshape(-1, 768) - g["small_feature_map"].reshape(-1, 768)).max(-1)
    assert d[ok].max() < 1e-4
    assert np.array_equal(o["positions"].reshape(-1, 3)[ok], g["small_positions"].reshape(-1, 3)[ok])
