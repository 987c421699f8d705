"""Float32 CPU restatement of the dense blocks on the path (torch primitives on CPU only).
TEST INFRASTRUCTURE -- see oracle/geometry.py header.

Weights are plain dicts keyed by the reference's state-dict names."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

T = torch.Tensor


def linear(x: T, sd: Dict[str, T], name: str) -> T:
    b = sd.get(name + ".bias")
    return F.linear(x, sd[name + ".weight"], b)


def layer_norm(x: T, sd: Dict[str, T], name: str, eps: float) -> T:
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[name + ".bias"], eps)


def mlp_ln_gelu(x: T, sd: Dict[str, T], name: str) -> T:
    """nn.Sequential(Linear, LayerNorm(eps 1e-5), GELU(erf), Linear)  (VLN-FF:139-143, 148-152,
    157-161; VLN-POL:83-111)."""
    h = linear(x, sd, name + ".0")
    h = layer_norm(h, sd, name + ".1", 1e-5)
    h = F.gelu(h)
    return linear(h, sd, name + ".3")


def mha(x: T, sd: Dict[str, T], name: str, n_head: int, key_mask: T | None = None) -> T:
    """nn.MultiheadAttention self-attention, batch_first, x (B,L,D); key_mask (B,L) True=valid."""
    B, L, D = x.shape
    qkv = F.linear(x, sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    hd = D // n_head
    sh = lambda t: t.view(B, L, n_head, hd).transpose(1, 2)
    q, k, v = sh(q), sh(k), sh(v)
    att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if key_mask is not None:
        att = att.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    att = torch.softmax(att, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, L, D)
    return F.linear(o, sd[name + ".out_proj.weight"], sd[name + ".out_proj.bias"])


def encoder_post_ln(x: T, sd: Dict[str, T], name: str, n_layers: int = 2, n_head: int = 12,
                    key_mask: T | None = None) -> T:
    """nn.TransformerEncoder(nn.TransformerEncoderLayer(768, 12, 3072, gelu, batch_first,
    norm_first=False), num_layers=2, norm=LayerNorm(eps=1e-12))   (VLN-FF:134-146).  Dropout is
    inactive (eval)."""
    for i in range(n_layers):
        p = f"{name}.layers.{i}"
        a = mha(x, sd, p + ".self_attn", n_head, key_mask)
        x = layer_norm(x + a, sd, p + ".norm1", 1e-5)
        h = linear(F.gelu(linear(x, sd, p + ".linear1")), sd, p + ".linear2")
        x = layer_norm(x + h, sd, p + ".norm2", 1e-5)
    return layer_norm(x, sd, name + ".norm", 1e-12)


def encode_set(tokens: T, cls: T, sd: Dict[str, T], name: str) -> T:
    """[CLS; tokens] -> encoder -> row 0.  tokens (n,768), cls (1,768) -> (1,768)."""
    seq = torch.cat([cls, tokens], dim=0).unsqueeze(0)
    return encoder_post_ln(seq, sd, name)[0, 0:1]


def tcnn_mlp(x: T, weights, act: str = "LeakyReLU", out_act: str = "None", store=None) -> T:
    """tinycudann CutlassMLP as used at PRE-FF:221-243: bias-free layers y = act(x W^T), LeakyReLU slope 0.01 (tiny-cuda-nn's
    `leaky_relu`), differentiable (plain torch ops) -- the float32 restatement the HIP forward / backward are checked against.
    store=torch.float16: every layer's output is stored in fp16 like tinycudann's (and the kernels'); the cast is differentiable
    (identity gradient), and the LeakyReLU slope of the backward pass is then decided on the STORED activation, as it is there."""
    h = x
    for i, w in enumerate(weights):
        h = F.linear(h, w)
        a = out_act if i == len(weights) - 1 else act
        if a == "LeakyReLU":
            h = F.leaky_relu(h, 0.01)
        if store is not None:
            h = h.to(store).float()
    return h
