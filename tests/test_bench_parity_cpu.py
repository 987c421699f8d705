"""CPU: the arithmetic of bench.py's `parity` object (hard band criterion, per-row figures, prefix-token-row distances) on synthetic
arrays -- the block itself runs on the GPU box behind the timed region."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_parity_block_hard_criterion_and_per_row():
    b = _bench()
    rng = np.random.default_rng(0)
    ref = rng.standard_normal((4, 1000)).astype(np.float32)
    lowp = ref + 0.02 * rng.standard_normal(ref.shape).astype(np.float32)
    near = ref + 0.9 * (lowp - ref) + 0.002 * rng.standard_normal(ref.shape).astype(np.float32)   # shares most of the lowp evaluation's noise
    far = ref + 0.05 * rng.standard_normal(ref.shape).astype(np.float32)
    p = b.parity_block(near, ref, True, lowp)
    assert p["within_band"] is True and p["within_band_vs_lowp"] and p["within_band_vs_f32"] and p["logits_rel_l2_vs_lowp"] < p["band"]
    indep = ref + 0.02 * rng.standard_normal(ref.shape).astype(np.float32)         # independent noise of the band's size: sqrt(2) from lowp
    assert b.parity_block(indep, ref, True, lowp)["within_band_vs_lowp"] is False
    assert len(p["logits_rel_l2_per_row"]) == 4 and p["logits_rel_l2_per_row_max"] == max(p["logits_rel_l2_per_row"])
    assert b.parity_block(far, ref, True, lowp)["within_band"] is False
    q = b.parity_block(near, ref, True)                                           # no lowp leg: the g19 band is quoted, nothing is asserted
    assert q["within_band"] is None and (q["band"] is None or q["band"] > 0)


def test_prefix_rows_parity_cuts_rows_by_kind():
    b = _bench()
    rng = np.random.default_rng(1)
    P, H = 8, 32
    counts = dict(Ni=[3, 0], Nz=[2, 1])
    lens = [2 + P + 3 + 2 + 5, 2 + P + 0 + 1 + 4]
    orc = torch.zeros(2, max(lens), H)
    rows = []
    for e, n in enumerate(lens):
        orc[e, :n] = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32))
        r = orc[e, :n].clone()
        r[2:2 + P] += 0.01 * torch.randn(P, H)                                     # patch rows perturbed, instance / zone rows only bf16-rounded
        rows.append(r)
    x = torch.cat(rows + [torch.zeros(3, H)], 0).to(torch.bfloat16)
    out = b.prefix_rows_parity((x, lens), orc, lens, counts, P=P)
    assert 5e-3 < out["rel_l2"]["patch"] < 2e-2
    assert abs(out["rel_l2"]["instance"] - out["bf16_store_floor"]["instance"]) < 1e-6 and out["rel_l2"]["instance"] < 3e-3
    assert out["rel_l2"]["zone"] < 3e-3 and out["north_star_1e3_met"] is False
