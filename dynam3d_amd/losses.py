"""Pre-training losses of the Pretrain `Feature_Fields.update_feature_fields(is_training=True)` (first slice):

  * `contrastive_loss`      PRE-FF:828-840: symmetric InfoNCE over a (n, n) similarity matrix, logit scale 10
  * `alignment_loss`        PRE-FF:1302-1330: the `sim_loss` assembly -- contrastive / 5 + (1 - cosine) on the instance features,
                            (1 - cosine) on the subspace-centred features (+1e-7 in the norm), the same for zones when image features
                            are given, the zone subspace term only when its target is not all zero
  * `segmentation_loss`     PRE-FF:1034-1046: class-balanced cross-entropy of the merge discriminator's SOFTMAXED scores (the reference
                            feeds probabilities to `F.cross_entropy`), first `min(#pos, #neg)` of each class in order

All are a few hundred rows by 768 columns: differentiable PyTorch expressions in float32 on the device the features live on (the
dense work of the training step is the encoders and the tcnn MLPs, whose backward is HIP: dynam3d_amd/tcnn.py)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def sim_matrix_cross_entropy(sim: torch.Tensor) -> torch.Tensor:
    return -torch.diag(F.log_softmax(sim, dim=-1)).mean()


def contrastive_loss(a: torch.Tensor, b: torch.Tensor, logit_scale: float = 10.0) -> torch.Tensor:
    sim = logit_scale * (a @ b.t())
    return sim_matrix_cross_entropy(sim) + sim_matrix_cross_entropy(sim.t())


def _unit(x, eps=0.0):
    return x / (torch.linalg.norm(x, dim=-1, keepdim=True) + eps)


def alignment_loss(pred_inst, tgt_inst, pred_inst_sub, tgt_inst_sub, pred_zone: Optional[torch.Tensor] = None, tgt_zone=None,
                   pred_zone_sub=None, tgt_zone_sub=None) -> torch.Tensor:
    # float16 targets (means of the float16 CLIP features) are normalised in float16 like the reference's (PRE-FF:1307, 1314), then promoted
    p, t = _unit(pred_inst), _unit(tgt_inst).float()
    loss = contrastive_loss(p, t) / 5.0 + (1.0 - (p * t).sum(-1)).mean()
    ps, ts = _unit(pred_inst_sub, 1e-7), _unit(tgt_inst_sub, 1e-7).float()
    loss = loss + (1.0 - (ps * ts).sum(-1)).mean()
    if pred_zone is not None:
        pz, tz = _unit(pred_zone), _unit(tgt_zone)
        loss = loss + contrastive_loss(pz, tz) / 5.0 + (1.0 - (pz * tz).sum(-1)).mean()
        if float(tgt_zone_sub.detach().sum()) != 0.0:
            pzs, tzs = _unit(pred_zone_sub), _unit(tgt_zone_sub)
            loss = loss + (1.0 - (pzs * tzs).sum(-1)).mean()
    return loss


def render_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """PRE-TR:1056-1075: rendered novel-view patch features `pred` (views, rays, 768) against the CLIP patch features of the view's image
    (2 x 2 average-pooled to the 12 x 12 ray grid): 2 x (1 - cosine) in the per-view mean-centred subspace, 5 x (1 - cosine), and the
    symmetric InfoNCE over all rays / 5.  Norms carry the reference's + 1e-5."""
    pred, target = pred.float(), target.float()
    ps, ts = pred - pred.mean(1, keepdim=True), target - target.mean(1, keepdim=True)
    ps, ts = _unit(ps, 1e-5), _unit(ts, 1e-5)
    loss = (1.0 - (ps * ts).sum(-1)).mean() * 2.0
    p, t = _unit(pred, 1e-5).reshape(-1, pred.shape[-1]), _unit(target, 1e-5).reshape(-1, target.shape[-1])
    loss = loss + (1.0 - (p * t).sum(-1)).mean() * 5.0
    return loss + contrastive_loss(p, t, 10.0) / 5.0


def segmentation_loss(merge_logits: torch.Tensor, merge_target: torch.Tensor):
    """merge_logits (..., 2), merge_target (...) in {0, 1}.  Returns None when one class is absent (the reference then skips it)."""
    score = torch.softmax(merge_logits, dim=-1).reshape(-1, 2)
    gt = merge_target.reshape(-1)
    n1, n0 = int((gt == 1).sum()), int((gt == 0).sum())
    if n1 == 0 or n0 == 0:
        return None
    m = min(n1, n0)
    s = torch.cat([score[gt == 1][:m], score[gt == 0][:m]], 0)
    g = torch.cat([gt[gt == 1][:m], gt[gt == 0][:m]], 0)
    return F.cross_entropy(s, g)
