"""Training branch of the novel-view renderer (rows a20-a23 under `loss.backward()`): the reference trains
`patch_to_nerf_position_embedding`, `aggregate_patch_to_nerf_encoder`, `nerf_encoder` and `nerf_decoder` through
`render_view_3d_patch` (Dynam3D_Pretrain/src_3dff/ss_trainer_3DFF.py "PRE-TR":880-892, 1056-1075; models/feature_fields.py "PRE-FF":
446-491, 494-625) against the CLIP patch features of the novel view's own image.

What is differentiated: the stored patches are constants there (host numpy arrays, PRE-FF:613), so the gradient stops at the gathered
neighbour features and their geometry; it flows through

    Linear(6, 768) + LayerNorm            of every neighbour's relative geometry            -> float32 MFMA GEMM + LN kernels (train_ops)
    fp16 add, 4 neighbours side by side   (identity gradient through the 16-bit store)
    Linear(3072, 768) + LayerNorm         `aggregate_patch_to_nerf_encoder`                 -> float32 MFMA GEMM + LN kernels
    tcnn encoder 768-768-768-769, +residual, tcnn decoder 768-768-768-768                   -> fp16 MFMA forward / backward (tcnn.py)
    softplus density -> alpha compositing over the ray -> L2 normalisation (`raw2feature`)  -> d3d_composite / d3d_composite_bwd

Everything in front of that (rays, 72 144-query KNN, importance top-8, gather) is the inference renderer's `front()` stage, unchanged.
`TrainableRenderer.render(...)` returns the differentiable (B, 144, 768) feature map; `losses.render_loss` is PRE-TR:1056-1075.
On CPU tensors (GPU-less tests) the same expressions run in PyTorch; the tcnn networks are then emulated with fp16 stores per layer."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import train_ops as TO
from .train_ff import lin

RENDER_KEYS = ("patch_to_nerf_position_embedding.0.weight", "patch_to_nerf_position_embedding.0.bias", "patch_to_nerf_position_embedding.1.weight",
               "patch_to_nerf_position_embedding.1.bias", "aggregate_patch_to_nerf_encoder.0.weight", "aggregate_patch_to_nerf_encoder.0.bias",
               "aggregate_patch_to_nerf_encoder.1.weight", "aggregate_patch_to_nerf_encoder.1.bias")


class _CpuMlp(torch.nn.Module):
    """PyTorch emulation of `tcnn.Network` for CPU tensors: bias-free layers, fp16 weights and activations, float32 accumulation, an fp16
    store per layer (identity gradient through the stores), LeakyReLU(0.01) -- the arithmetic model of tcnn.py, differentiable."""

    def __init__(self, weights, out_act, round16=True):
        super().__init__()
        self.layers = torch.nn.ParameterList([torch.nn.Parameter(w.detach().float().clone()) for w in weights])
        self.out_act = out_act
        self.R = TO.RoundFp16.apply if round16 else (lambda t: t)

    def forward(self, x):
        R = self.R
        h = R(x)
        n = len(self.layers)
        for i, w in enumerate(self.layers):
            h = F.linear(h, R(w))
            if i < n - 1 or self.out_act == "LeakyReLU":
                h = F.leaky_relu(h, 0.01)
            h = R(h)
        return h


class TrainableRenderer(torch.nn.Module):
    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda", width: int = 768, k: int = 4, fp16_stores: bool = True):
        """`fp16_stores=False` (CPU tensors only): no 16-bit stores anywhere -- the same graph in plain float32, for tests that check the
        graph's wiring against the float64 oracle without the rounding-boundary noise of an fp16 network."""
        super().__init__()
        self.dev, self.width, self.k = torch.device(device), width, k
        if not fp16_stores and self.dev.type == "cuda":
            raise ValueError("the HIP tcnn networks store fp16 per layer: fp16_stores=False exists for the CPU arithmetic model only")
        self.R = TO.RoundFp16.apply if fp16_stores else (lambda t: t)
        self.names = list(RENDER_KEYS)
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(sd[n].detach().to(self.dev, torch.float32).clone()) for n in self.names])
        enc_w = [sd[f"nerf_encoder.layers.{i}.weight"] for i in range(3)]
        dec_w = [sd[f"nerf_decoder.layers.{i}.weight"] for i in range(3)]
        if self.dev.type == "cuda":
            from .tcnn import Network
            cfg = lambda out_act: {"otype": "CutlassMLP", "activation": "LeakyReLU", "output_activation": out_act, "n_neurons": width, "n_hidden_layers": 2}
            self.nerf_encoder = Network(width, width + 1, cfg("LeakyReLU"), enc_w, device)
            self.nerf_decoder = Network(width, width, cfg("None"), dec_w, device)
            for m in (self.nerf_encoder, self.nerf_decoder):
                m.float32_grad_io = True
        else:
            self.nerf_encoder, self.nerf_decoder = _CpuMlp(enc_w, "LeakyReLU", fp16_stores), _CpuMlp(dec_w, "None", fp16_stores)

    @property
    def w(self) -> Dict[str, torch.Tensor]:
        return dict(zip(self.names, self.plist))

    def layer_weights(self) -> Dict[str, torch.Tensor]:
        """Every trainable tensor under the checkpoint's per-layer names (the tcnn networks as [out, in] matrices)."""
        out = {k: p.detach() for k, p in zip(self.names, self.plist)}
        for name, net in (("nerf_encoder", self.nerf_encoder), ("nerf_decoder", self.nerf_decoder)):
            ws = net.layers_from_flat(net.params.detach()) if hasattr(net, "layers_from_flat") else [p.detach() for p in net.layers]
            for i, w_ in enumerate(ws):
                out[f"{name}.layers.{i}.weight"] = w_
        return out

    def layer_grads(self) -> Dict[str, Optional[torch.Tensor]]:
        out = {k: p.grad for k, p in zip(self.names, self.plist)}
        for name, net in (("nerf_encoder", self.nerf_encoder), ("nerf_decoder", self.nerf_decoder)):
            if hasattr(net, "layers_from_flat"):
                gs = [None] * 3 if net.params.grad is None else net.layers_from_flat(net.params.grad)
            else:
                gs = [p.grad for p in net.layers]
            for i, g in enumerate(gs):
                out[f"{name}.layers.{i}.weight"] = g
        return out

    @torch.enable_grad()
    def networks(self, feat16: torch.Tensor, geom6: torch.Tensor, rel_dist16: torch.Tensor, topk: torch.Tensor, n_samples: int):
        """feat16 (n * S, K * 768) fp16 gathered neighbour features, geom6 (n * S * K, 6) float32 -> (fmap (n, 768), depth (n,), debug dict)."""
        w, W_, K, R = self.w, self.width, self.k, self.R
        pos = TO.layer_norm(lin(geom6, w["patch_to_nerf_position_embedding.0.weight"], w["patch_to_nerf_position_embedding.0.bias"]),
                            w["patch_to_nerf_position_embedding.1.weight"], w["patch_to_nerf_position_embedding.1.bias"], 1e-12)      # PRE-FF:481
        x_in = R(feat16.float().view(-1, W_) + R(pos)).view(-1, K * W_)                               # fp16 add (PRE-FF:479-483)
        x = TO.layer_norm(lin(x_in, w["aggregate_patch_to_nerf_encoder.0.weight"], w["aggregate_patch_to_nerf_encoder.0.bias"]),
                          w["aggregate_patch_to_nerf_encoder.1.weight"], w["aggregate_patch_to_nerf_encoder.1.bias"], 1e-12)           # PRE-FF:483
        x = R(x)                                                                                                                        # `sample_input` is fp16
        enc = self.nerf_encoder(x)                                                                                                      # PRE-FF:484
        dens = enc[:, W_]
        y = R(enc[:, :W_] + x)                                                                                                          # residual, PRE-FF:487
        out = self.nerf_decoder(y)                                                                                                      # PRE-FF:488
        fmap, depth = TO.composite(out, dens, rel_dist16, topk, n_samples)                                                              # PRE-FF:446-474
        return fmap, depth, dict(pos=pos, x=x, enc=enc, out=out, dens=dens)

    def render(self, ff, batch_position, batch_heading):
        """Differentiable `render_view_3d_patch` of the feature field `ff` (habitat mode) -> (B, 144, 768) unit-norm features.  The front
        stage (rays / KNN / top-8 / gather) is the inference renderer's, without gradients."""
        if ff._renderer is None:
            from .render import FieldRenderer
            ff._renderer = FieldRenderer(ff._render_sd, ff.device)
        r = ff._renderer
        st = ff.state
        n_rows = [st.count(e, st.ROWS) for e in range(ff.batch_size)]
        with torch.no_grad():
            fr = r.front(ff.pools, ff.slots, n_rows, batch_position, batch_heading, ff.ops, want_geom6=True, raw_features=True)
        fmap, _depth, dbg = self.networks(fr["s16"], fr["geom6"], fr["rel_dist16"], fr["topk"], r.N)
        self.last = dict(front=fr, **dbg)
        return fmap.view(fr["B"], r.R, self.width)
