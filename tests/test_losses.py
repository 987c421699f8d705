"""CPU: Pretrain losses (first slice).  oracle/losses_ref.py against golden g15 -- the reference's own
`contrastive_loss` / `sim_matrix_cross_entropy` executed on seeded pairs, values AND autograd gradients -- and the product's
dynam3d_amd/losses.py (values and gradients through torch autograd) against the oracle on the full `sim_loss` assembly."""
import numpy as np
import torch

from oracle import losses_ref as LR
from tests.golden_io import load
from dynam3d_amd import losses as L


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_oracle_contrastive_loss_matches_reference_golden():
    g = load("g15_losses.npz")
    for i in range(int(g["n"])):
        a, b = g[f"a_{i}"], g[f"b_{i}"]
        loss, ga, gb = LR.contrastive_loss(a, b)
        assert abs(loss - float(g[f"loss_{i}"])) < 2e-6 * max(1.0, abs(loss))
        if a.shape[0] > 1:
            assert rel(ga, g[f"ga_{i}"]) < 1e-4 and rel(gb, g[f"gb_{i}"]) < 1e-4
        x, _ = LR.sim_matrix_cross_entropy(10.0 * a.astype(np.float64) @ b.astype(np.float64).T)
        assert abs(x - float(g[f"xent_{i}"])) < 2e-6 * max(1.0, abs(x))
        # product, float32 torch
        ta, tb = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
        lt = L.contrastive_loss(ta, tb)
        lt.backward()
        assert abs(float(lt) - float(g[f"loss_{i}"])) < 2e-6 * max(1.0, abs(loss))
        if a.shape[0] > 1:
            assert rel(ta.grad.numpy(), g[f"ga_{i}"]) < 1e-5 and rel(tb.grad.numpy(), g[f"gb_{i}"]) < 1e-5


def test_alignment_and_segmentation_losses_match_oracle():
    rng = np.random.default_rng(7)
    n, nz, d = 37, 9, 768
    mk = lambda r: rng.standard_normal((r, d)).astype(np.float32)
    args = dict(pred_inst=mk(n), tgt_inst=mk(n), pred_inst_sub=mk(n), tgt_inst_sub=mk(n), pred_zone=mk(nz), tgt_zone=mk(nz),
                pred_zone_sub=mk(nz), tgt_zone_sub=mk(nz))
    for with_zone, zero_sub in ((True, False), (True, True), (False, False)):
        kw = dict(args)
        if zero_sub:
            kw["tgt_zone_sub"] = np.zeros_like(kw["tgt_zone_sub"])
        if not with_zone:
            kw = {k: v for k, v in kw.items() if "zone" not in k}
        ref, grads = LR.alignment_loss(**kw)
        t = {k: torch.from_numpy(v).requires_grad_(k.startswith("pred")) for k, v in kw.items()}
        got = L.alignment_loss(**t)
        got.backward()
        assert abs(float(got) - ref) < 1e-5 * max(1.0, abs(ref)), (with_zone, zero_sub, float(got), ref)
        for k, gref in grads.items():
            assert rel(t[k].grad.numpy(), gref) < 1e-4, (k, with_zone, zero_sub)
        if with_zone and zero_sub:
            assert t["pred_zone_sub"].grad is None                  # the term is skipped (PRE-FF:1325)
    logits = rng.standard_normal((11, 4, 2)).astype(np.float32)
    tgt = (rng.random((11, 4)) < 0.3).astype(np.int64)
    ref = LR.segmentation_loss(logits, tgt)
    got = L.segmentation_loss(torch.from_numpy(logits), torch.from_numpy(tgt))
    assert abs(float(got) - ref) < 1e-6
    assert L.segmentation_loss(torch.from_numpy(logits), torch.zeros(11, 4, dtype=torch.int64)) is None
    assert LR.segmentation_loss(logits, np.zeros((11, 4), np.int64)) is None
