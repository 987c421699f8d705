"""Float64 restatement of ONE pre-training forward/backward of the feature field.  TEST INFRASTRUCTURE (see
oracle/geometry.py): only tests may import it.

Follows the reference's `update_feature_fields(is_training=True)` literally -- one encoder call per 2D segment (PRE-FF:940-966), the
alignment targets of PRE-FF:969-974, the frame-as-one-zone encoding of PRE-FF:989-1008, the merge discriminator on [ft_3d, ft_2d,
position offset] (PRE-FF:1019-1027) with the class-balanced cross-entropy of PRE-FF:1034-1046, and the `sim_loss` assembly of
PRE-FF:1302-1330 -- in float64 with torch autograd, using the oracle's own module restatements (oracle/nnref.py).  Returns the loss and
every parameter's gradient.

The per-view records (`views`) are built FROM RAW INPUTS by the oracle's own memory state machine --
`oracle.ff_oracle.FeatureFieldsOracle.update_feature_fields(train=...)`: segment grouping, GT labels, proposals, ground-truth merge
targets --, and the whole chain is pinned by golden g21 (tests/golden/gen_golden_train.py): the REFERENCE's
`update_feature_fields(is_training=True)` executed on the CPU (oracle/ref_harness.RefTrainingRun), its losses, its gradients, the
nearest-GT-point indices, GT ids, merge targets and zone member lists.  (tests/test_train_ff.py additionally feeds it the records the
PRODUCT exported, `FFTrainer.debug`, to check the product's gradients element by element.)

Targets are means of float16 CLIP features evaluated IN float16 like the reference does (`patch_fts[...].mean(0)` on a half tensor,
PRE-FF:969-972): torch's own half mean, then float64."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import nnref as NN


def _ce_sim(sim):
    return -torch.diag(F.log_softmax(sim, dim=-1)).mean()


def _contrastive(a, b):
    sim = 10.0 * (a @ b.t())
    return _ce_sim(sim) + _ce_sim(sim.t())


def _unit(x, eps=0.0):
    return x / (torch.linalg.norm(x, dim=-1, keepdim=True) + eps)


def training_loss_and_grads(sd: Dict[str, torch.Tensor], views: List[dict]):
    """views[i]: tok_fts (T,768), geom7 (T,7), lens (G,), cen (G,3), env_of_group (G,), B, P, img_ix / img_mean (B,768) or None,
    pairs: None or dict(f3 (n,768), p3 (n,3), g (n,), target (n,), pe (n,)).  -> (loss, sim_loss, segm_loss or None, {name: grad})."""
    w = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    npy = lambda t: np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t)
    d = lambda t: torch.as_tensor(npy(t)).double()
    h16 = lambda t: torch.as_tensor(npy(t)).to(torch.float16)
    pi, ti, pis, tis, pz, tz, pzs, tzs, segm = [], [], [], [], [], [], [], [], []
    ce_records = []
    for v in views:
        tok, g7, cen = d(v["tok_fts"]), d(v["geom7"]), d(v["cen"])
        tok16 = h16(v["tok_fts"])                                               # (the CLIP grid features ARE float16 values, PRE-FF:845)
        lens = np.asarray(v["lens"], np.int64)
        env = np.asarray(v["env_of_group"], np.int64)
        B, P = int(v["B"]), int(v["P"])
        off = np.concatenate([[0], np.cumsum(lens)])
        frame_mean = tok16.view(B, P, -1).mean(1).double()                      # patch_fts.mean(0, keepdim=True) in float16 (PRE-FF:971)
        preds = []
        for gi in range(len(lens)):                                              # one encoder call per segment (PRE-FF:940-966)
            t = tok[off[gi]:off[gi + 1]]
            emb = t + NN.mlp_ln_gelu(g7[off[gi]:off[gi + 1]], w, "patch_to_instance_position_embedding")
            ft = NN.encode_set(emb, w["aggregate_patch_to_instance_embedding"], w, "aggregate_patch_to_instance_encoder")
            preds.append(ft)
            tmean = tok16[off[gi]:off[gi + 1]].mean(0, keepdim=True)             # float16 mean (PRE-FF:969)
            pi.append(ft); ti.append(tmean)                                      # targets stay float16 up to their normalisation, like the reference's
            # (half mean) - (half frame mean) is a half subtraction in the reference (PRE-FF:971)
            pis.append(ft - frame_mean[env[gi]:env[gi] + 1]); tis.append(tmean - tok16.view(B, P, -1).mean(1)[env[gi]:env[gi] + 1])
        pred = torch.cat(preds, 0)
        if v.get("img_ix") is not None:
            img_ix, img_mean = d(v["img_ix"]), d(v["img_mean"])
            for b in range(B):                                                   # PRE-FF:989-1008
                sel = np.nonzero(env == b)[0]
                ip = cen[sel]
                pe_ = torch.cat([ip - ip.mean(0, keepdim=True), torch.sqrt((ip * ip).sum(-1)).unsqueeze(-1)], -1)
                st = pred[sel] + NN.mlp_ln_gelu(pe_, w, "instance_to_zone_position_embedding")
                z = NN.encode_set(st, w["aggregate_instance_to_zone_embedding"], w, "aggregate_instance_to_zone_encoder")
                pz.append(z); tz.append(img_ix[b:b + 1])
                pzs.append(z - img_mean[b:b + 1]); tzs.append(img_ix[b:b + 1] - img_mean[b:b + 1])
        pr = v.get("pairs")
        if pr is not None:
            g = torch.as_tensor(npy(pr["g"]), dtype=torch.long)
            x = torch.cat([d(pr["f3"]), pred[g], cen[g] - d(pr["p3"])], -1)
            score = torch.softmax(NN.mlp_ln_gelu(x, w, "instance_merge_discriminator"), -1)
            tgt = torch.as_tensor(npy(pr["target"]), dtype=torch.long)
            pe = npy(pr["pe"])
            for b in range(B):                                                   # PRE-FF:1034-1046, per (environment, view)
                m = torch.from_numpy(pe == b)
                s_, t_ = score[m], tgt[m]
                n1, n0 = int((t_ == 1).sum()), int((t_ == 0).sum())
                if n1 and n0:
                    k = min(n1, n0)
                    sc, tg = torch.cat([s_[t_ == 1][:k], s_[t_ == 0][:k]]), torch.cat([t_[t_ == 1][:k], t_[t_ == 0][:k]])
                    ce_records.append((sc.detach().numpy().copy(), tg.numpy().copy()))
                    segm.append(F.cross_entropy(sc, tg))
    cat = lambda xs: torch.cat(xs, 0)
    # the float16 targets are normalised IN float16 (PRE-FF:1307, 1314: half / half-norm; the 1e-7 is absorbed by the half sum)
    p, t = _unit(cat(pi)), _unit(cat(ti)).double()
    sim = _contrastive(p, t) / 5.0 + (1.0 - (p * t).sum(-1)).mean()
    ps, ts = _unit(cat(pis), 1e-7), _unit(cat(tis), 1e-7).double()
    sim = sim + (1.0 - (ps * ts).sum(-1)).mean()
    if pz:
        a, b_ = _unit(cat(pz)), _unit(cat(tz))
        sim = sim + _contrastive(a, b_) / 5.0 + (1.0 - (a * b_).sum(-1)).mean()
        if float(cat(tzs).sum()) != 0.0:
            a, b_ = _unit(cat(pzs)), _unit(cat(tzs))
            sim = sim + (1.0 - (a * b_).sum(-1)).mean()
    seg = torch.stack(segm).mean() if segm else None
    loss = sim if seg is None else sim + seg
    loss.backward()
    training_loss_and_grads.last_ce_records = ce_records        # the balanced (score, target) sets of every cross-entropy term, in call order
    return float(loss.detach()), float(sim.detach()), None if seg is None else float(seg.detach()), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in w.items()}
