"""CPU restatement (numpy, float64) of the Pretrain losses with analytic gradients.  TEST INFRASTRUCTURE (see oracle/geometry.py).

  contrastive_loss      <- PRE-FF:828-840 (`sim_matrix_cross_entropy`, `contrastive_loss`)
  alignment_loss        <- PRE-FF:1302-1330 (the `sim_loss` assembly)
  segmentation_loss     <- PRE-FF:1034-1046
Pinned by tests/golden/g15_losses.npz: `contrastive_loss` / `sim_matrix_cross_entropy` are EXECUTED from the reference's Pretrain
class on seeded feature pairs (tests/golden/gen_golden_losses.py); the assembly lines cannot be executed on a CPU (`is_training=True`
needs CUDA autocast) and are restated line by line."""
from __future__ import annotations

import numpy as np


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def sim_matrix_cross_entropy(sim):
    """-> (loss, dloss/dsim)"""
    n = sim.shape[0]
    lp = _log_softmax(sim)
    g = np.exp(lp) / n
    g[np.arange(n), np.arange(n)] -= 1.0 / n
    return -np.diag(lp).mean(), g


def contrastive_loss(a, b, logit_scale=10.0):
    """-> (loss, dl/da, dl/db)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    sim = logit_scale * (a @ b.T)
    l1, g1 = sim_matrix_cross_entropy(sim)
    l2, g2 = sim_matrix_cross_entropy(sim.T)
    g = g1 + g2.T
    return l1 + l2, logit_scale * (g @ b), logit_scale * (g.T @ a)


def _unit(x, eps=0.0):
    """-> (y = x / (|x| + eps), vjp(dy) -> dx)"""
    n = np.linalg.norm(x, axis=-1, keepdims=True)
    d = n + eps
    y = x / d

    def vjp(dy):
        return dy / d - x * ((dy * x).sum(-1, keepdims=True) / (d * d * np.maximum(n, 1e-300)))
    return y, vjp


def _cos_term(p, t):
    """(1 - sum(p*t, -1)).mean() -> (loss, dl/dp)"""
    return (1.0 - (p * t).sum(-1)).mean(), -t / p.shape[0]


def alignment_loss(pred_inst, tgt_inst, pred_inst_sub, tgt_inst_sub, pred_zone=None, tgt_zone=None, pred_zone_sub=None, tgt_zone_sub=None):
    """-> (loss, dict of gradients w.r.t. the PREDICTED features)"""
    f64 = lambda x: None if x is None else np.asarray(x, np.float64)
    pred_inst, tgt_inst, pred_inst_sub, tgt_inst_sub = map(f64, (pred_inst, tgt_inst, pred_inst_sub, tgt_inst_sub))
    grads = {}
    p, vp = _unit(pred_inst)
    t, _ = _unit(tgt_inst)
    lc, gp, _ = contrastive_loss(p, t)
    l2, g2 = _cos_term(p, t)
    loss = lc / 5.0 + l2
    grads["pred_inst"] = vp(gp / 5.0 + g2)
    ps, vps = _unit(pred_inst_sub, 1e-7)
    ts, _ = _unit(tgt_inst_sub, 1e-7)
    l3, g3 = _cos_term(ps, ts)
    loss += l3
    grads["pred_inst_sub"] = vps(g3)
    if pred_zone is not None:
        pred_zone, tgt_zone, pred_zone_sub, tgt_zone_sub = map(f64, (pred_zone, tgt_zone, pred_zone_sub, tgt_zone_sub))
        pz, vpz = _unit(pred_zone)
        tz, _ = _unit(tgt_zone)
        lc, gp, _ = contrastive_loss(pz, tz)
        l2, g2 = _cos_term(pz, tz)
        loss += lc / 5.0 + l2
        grads["pred_zone"] = vpz(gp / 5.0 + g2)
        if tgt_zone_sub.sum() != 0:
            pzs, vpzs = _unit(pred_zone_sub)
            tzs, _ = _unit(tgt_zone_sub)
            l4, g4 = _cos_term(pzs, tzs)
            loss += l4
            grads["pred_zone_sub"] = vpzs(g4)
    return loss, grads


def segmentation_loss(merge_logits, merge_target):
    """-> loss or None.  Cross-entropy applied to SOFTMAXED scores, as the reference does."""
    z = np.asarray(merge_logits, np.float64).reshape(-1, 2)
    gt = np.asarray(merge_target).reshape(-1)
    n1, n0 = int((gt == 1).sum()), int((gt == 0).sum())
    if n1 == 0 or n0 == 0:
        return None
    m = min(n1, n0)
    score = np.exp(_log_softmax(z))
    s = np.concatenate([score[gt == 1][:m], score[gt == 0][:m]])
    g = np.concatenate([gt[gt == 1][:m], gt[gt == 0][:m]])
    lp = _log_softmax(s)
    return -lp[np.arange(len(g)), g].mean()
