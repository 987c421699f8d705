"""Float64 restatement of the renderer's training forward / backward (`patch_to_nerf_encode` PRE-FF:477-491, `raw2feature`
PRE-FF:446-474, the render losses PRE-TR:1056-1075) on the inputs the product's front stage exported.  TEST INFRASTRUCTURE (see
oracle/geometry.py): only tests may import it.

Written the reference's way -- `nn.Linear` / BertLayerNorm(eps 1e-12) expressions, the tcnn networks as bias-free LeakyReLU(0.01) layers
(PRE-FF:221-243), `raw2feature` literally: softplus, scatter of the 8 sample densities into the 501-bin ray, alpha, cumprod, gather,
weighted sum, L2 normalisation -- in float64 with torch autograd.  The 16-bit STORES of the reference's fp16 arithmetic (tcnn layer outputs,
the fp16 add of features and position embedding, the fp16 `sample_input`) are applied as round-to-fp16 with an identity gradient, the same
places the kernels store 16 bit (dynam3d_amd/train_render.py), so that value and gradient are comparable at float32 accuracy instead of fp16's."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


class _R16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _mlp(x, ws, out_act, R=_R16.apply):
    h = R(x)
    for i, w in enumerate(ws):
        h = F.linear(h, R(w))
        if i < len(ws) - 1 or out_act == "LeakyReLU":
            h = F.leaky_relu(h, 0.01)
        h = R(h)
    return h


def _raw2feature(sample_feature, sample_density, rel_dist, topk_inds):
    """PRE-FF:446-474 on (rays, 8, 768) / (rays, 8) / (rays, 501) / (rays, 8).  The `+ 1e-10` the reference adds to every bin's transmittance
    factor is below float32 resolution next to 1.0 (it runs in 16 / 32 bit): applied to the bins that carry density only."""
    sd = F.softplus(sample_density)
    dists = torch.abs(rel_dist[..., 1:] - rel_dist[..., :-1])
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    density = torch.zeros_like(rel_dist).scatter(1, topk_inds, sd)
    alpha = 1.0 - torch.exp(-F.relu(density) * dists)
    occupied = torch.zeros_like(rel_dist).scatter(1, topk_inds, torch.ones_like(sd))
    factor = 1.0 - alpha + 1e-10 * occupied
    weights = alpha * torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), factor], -1), -1)[:, :-1]
    sw = torch.gather(weights, 1, topk_inds)
    fmap = torch.sum(sw[..., None] * sample_feature, -2)
    return fmap / torch.clamp(torch.linalg.norm(fmap, dim=-1, keepdim=True), min=1e-7)


def _ce(sim):
    return -torch.diag(F.log_softmax(sim, dim=-1)).mean()


def render_loss(pred, target):
    """PRE-TR:1056-1075."""
    unit = lambda x: x / (torch.linalg.norm(x, dim=-1, keepdim=True) + 1e-5)
    ps, ts = unit(pred - pred.mean(1, keepdim=True)), unit(target - target.mean(1, keepdim=True))
    loss = (1.0 - (ps * ts).sum(-1)).mean() * 2.0
    p, t = unit(pred).reshape(-1, pred.shape[-1]), unit(target).reshape(-1, target.shape[-1])
    loss = loss + (1.0 - (p * t).sum(-1)).mean() * 5.0
    sim = 10.0 * (p @ t.t())
    return loss + (_ce(sim) + _ce(sim.t())) / 5.0


def render_loss_and_grads(sd: Dict[str, torch.Tensor], feat16, geom6, rel_dist16, topk, n_samples: int, target, n_views: int, k: int = 4,
                          width: int = 768, stores: bool = True):
    """sd: the renderer's parameters (per-layer tcnn matrices `nerf_*.layers.i.weight`); feat16 (n * S, k * width) gathered neighbour
    features, geom6 (n * S * k, 6), rel_dist16 (n_samples,), topk (n, S), target (n_views, n / n_views, width).
    `stores=False`: no 16-bit stores (plain float64): an fp16 network's LeakyReLU slopes are decided on stored activations, and two evaluations
    that round a few activations to different fp16 neighbours choose a few slopes differently -- ~1e-2 of gradient noise at these sizes that has
    nothing to do with correctness; the store-free graph is the one that can be compared tightly.
    -> (loss, feature map (n, width), {name: gradient})."""
    R = _R16.apply if stores else (lambda t: t)
    d = lambda t: torch.as_tensor(np.asarray(t.detach().cpu() if isinstance(t, torch.Tensor) else t)).double()
    w = {k_: v.detach().cpu().double().clone().requires_grad_(True) for k_, v in sd.items()}
    ln = lambda x, name: F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], 1e-12)
    feat, g6 = d(feat16.float()), d(geom6)
    pos = ln(F.linear(g6, w["patch_to_nerf_position_embedding.0.weight"], w["patch_to_nerf_position_embedding.0.bias"]), "patch_to_nerf_position_embedding.1")
    x_in = R(feat.view(-1, width) + R(pos)).view(-1, k * width)
    x = R(ln(F.linear(x_in, w["aggregate_patch_to_nerf_encoder.0.weight"], w["aggregate_patch_to_nerf_encoder.0.bias"]), "aggregate_patch_to_nerf_encoder.1"))
    enc = _mlp(x, [w[f"nerf_encoder.layers.{i}.weight"] for i in range(3)], "LeakyReLU", R)
    dens = enc[:, width]
    y = R(enc[:, :width] + x)
    out = _mlp(y, [w[f"nerf_decoder.layers.{i}.weight"] for i in range(3)], "None", R)
    tk = torch.as_tensor(np.asarray(topk.cpu())).long()
    n, S = tk.shape
    rd = d(rel_dist16)[None].expand(n, n_samples)
    fmap = _raw2feature(out.view(n, S, width), dens.view(n, S), rd, tk)
    loss = render_loss(fmap.view(n_views, -1, width), d(target))
    loss.backward()
    return float(loss.detach()), fmap.detach(), {k_: (v.grad if v.grad is not None else torch.zeros_like(v)) for k_, v in w.items()}
