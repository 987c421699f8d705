"""Float32 CPU restatement of the dense towers on the step path.  TEST INFRASTRUCTURE (see
oracle/geometry.py header).

  * clip_vit_forward      <- encoders/clip/model.py:162-238 (QuickGELU, ResidualAttentionBlock, VisionTransformer)
                             + encoders/resnet_encoders.py:267-284 (CLIPEncoder preprocessing)
  * llava_image_features  <- VLN-POL:448-452 `llava.get_image_features(layer -2, 'default')`; the HF CLIP vision
                             model / projector arithmetic is third-party (`transformers==4.46.0`, not vendored):
                             pinned against the installed transformers' CLIPVisionModel on seeded small configs
  * phi3_prefill_logits   <- VLN-POL:463 (llava.generate -> first-token logits); Phi-3 arithmetic pinned against the
                             installed transformers' Phi3ForCausalLM on seeded small configs
  * prefix MLPs / splice  <- VLN-POL:432-461
Pinning scripts: tests/golden/gen_golden_dense.py (goldens g5, g8, g9).

`lowp=torch.float16 / torch.bfloat16` evaluates the SAME restatement the way the reference evaluates its modules in
its own dtypes (OpenAI CLIP after `convert_weights` in fp16, clip/model.py:373-395; llava / Phi-3 with
`torch_dtype=torch.bfloat16`, VLN-POL:125): float32 arithmetic inside a module, a 16-bit store at every module
output (Linear, activation, residual add, norm, rotary products, softmax @ V), parameters rounded to the dtype
they are held in.  Pinned by tests/golden/gen_golden_lowp.py (g11-g14: the real modules run in those dtypes).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import nnref as NN

T = torch.Tensor
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_rgb(rgb_u8: np.ndarray, size: int = 336) -> T:
    """(B,h,w,3) uint8 -> (B,3,size,size) f32.  torchvision Resize(bicubic) on a uint8 tensor interpolates in
    float and rounds back to uint8; ConvertImageDtype(float) divides by 255; Normalize (resnet_encoders.py:267-271)."""
    x = torch.from_numpy(np.ascontiguousarray(rgb_u8)).permute(0, 3, 1, 2).float()
    if x.shape[-1] != size or x.shape[-2] != size:
        x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=False).round().clamp(0, 255)
    x = x / 255.0
    m, s = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (x - m) / s


def _rounder(lowp):
    """R(x): store-and-reload in the module dtype (identity for float32)."""
    if lowp is None or lowp == torch.float32:
        return lambda x: x
    return lambda x: x.to(lowp).float()


def _vit_blocks(x: T, get, n_layers: int, heads: int, lowp=None, ln_lowp: bool = False) -> T:
    """Pre-LN residual blocks with QuickGELU MLP; `get(i, name)` returns the (fused-qkv) tensors.
    lowp: 16-bit module dtype (see module docstring); ln_lowp: LayerNorm parameters are held in that dtype too
    (HF `model.to(bfloat16)`) or stay float32 (OpenAI `convert_weights` skips LayerNorm)."""
    B, L, W = x.shape
    hd = W // heads
    R = _rounder(lowp)
    Rn = R if ln_lowp else (lambda t: t)
    for i in range(n_layers):
        h = R(F.layer_norm(x, (W,), Rn(get(i, "ln1_w")), Rn(get(i, "ln1_b")), 1e-5))
        qkv = R(F.linear(h, R(get(i, "qkv_w")), R(get(i, "qkv_b")))).view(B, L, 3, heads, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)        # fused attention: float32 scores / softmax,
        a = R(R(att) @ v).transpose(1, 2).reshape(B, L, W)                             # probabilities and the output stored 16-bit
        x = R(x + R(F.linear(a, R(get(i, "out_w")), R(get(i, "out_b")))))
        h = R(F.layer_norm(x, (W,), Rn(get(i, "ln2_w")), Rn(get(i, "ln2_b")), 1e-5))
        h = R(F.linear(h, R(get(i, "fc1_w")), R(get(i, "fc1_b"))))
        h = R(h * R(torch.sigmoid(R(1.702 * h))))                                      # x * sigmoid(1.702 * x), three stores
        x = R(x + R(F.linear(h, R(get(i, "fc2_w")), R(get(i, "fc2_b")))))
    return x


def _embed(pixels: T, patch_w: T, cls: T, pos: T, patch: int, lowp=None) -> T:
    R = _rounder(lowp)
    x = R(F.conv2d(R(pixels), R(patch_w), stride=patch))
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([R(cls).view(1, 1, -1).expand(x.shape[0], 1, -1), x], 1)
    return R(x + R(pos))


def clip_vit_forward(pixels: T, sd: Dict[str, T], layers: int, heads: int, patch: int = 14, lowp=None):
    p = "visual."
    names = dict(ln1_w="ln_1.weight", ln1_b="ln_1.bias", ln2_w="ln_2.weight", ln2_b="ln_2.bias", qkv_w="attn.in_proj_weight",
                 qkv_b="attn.in_proj_bias", out_w="attn.out_proj.weight", out_b="attn.out_proj.bias", fc1_w="mlp.c_fc.weight",
                 fc1_b="mlp.c_fc.bias", fc2_w="mlp.c_proj.weight", fc2_b="mlp.c_proj.bias")
    get = lambda i, n: sd[f"{p}transformer.resblocks.{i}.{names[n]}"].float()
    R = _rounder(lowp)
    x = _embed(pixels, sd[p + "conv1.weight"].float(), sd[p + "class_embedding"].float(), sd[p + "positional_embedding"].float(), patch, lowp)
    W = x.shape[-1]
    x = R(F.layer_norm(x, (W,), sd[p + "ln_pre.weight"].float(), sd[p + "ln_pre.bias"].float(), 1e-5))     # LayerNorm stays float32 (model.py:153-159)
    x = _vit_blocks(x, get, layers, heads, lowp, ln_lowp=False)
    x = R(F.layer_norm(x, (W,), sd[p + "ln_post.weight"].float(), sd[p + "ln_post.bias"].float(), 1e-5))
    y = R(x @ R(sd[p + "proj"].float()))
    return y[:, 0], y[:, 1:]


def llava_image_features(pixels: T, sd: Dict[str, T], layers: int, heads: int, patch: int = 14, feature_layer: int = -2, lowp=None) -> T:
    v = "vision_tower.vision_model."
    R = _rounder(lowp)

    def get(i, n):
        q = f"{v}encoder.layers.{i}."
        if n == "qkv_w":
            return torch.cat([sd[q + f"self_attn.{x}_proj.weight"] for x in "qkv"], 0).float()
        if n == "qkv_b":
            return torch.cat([sd[q + f"self_attn.{x}_proj.bias"] for x in "qkv"], 0).float()
        m = dict(ln1_w="layer_norm1.weight", ln1_b="layer_norm1.bias", ln2_w="layer_norm2.weight", ln2_b="layer_norm2.bias",
                 out_w="self_attn.out_proj.weight", out_b="self_attn.out_proj.bias", fc1_w="mlp.fc1.weight", fc1_b="mlp.fc1.bias",
                 fc2_w="mlp.fc2.weight", fc2_b="mlp.fc2.bias")
        return sd[q + m[n]].float()

    x = _embed(pixels, sd[v + "embeddings.patch_embedding.weight"].float(), sd[v + "embeddings.class_embedding"].float(),
               sd[v + "embeddings.position_embedding.weight"].float(), patch, lowp)
    W = x.shape[-1]
    x = R(F.layer_norm(x, (W,), R(sd[v + "pre_layrnorm.weight"].float()), R(sd[v + "pre_layrnorm.bias"].float()), 1e-5))
    x = _vit_blocks(x, get, layers + 1 + feature_layer, heads, lowp, ln_lowp=True)[:, 1:]
    h = R(F.gelu(R(F.linear(x, R(sd["multi_modal_projector.linear_1.weight"].float()), R(sd["multi_modal_projector.linear_1.bias"].float())))))
    return R(F.linear(h, R(sd["multi_modal_projector.linear_2.weight"].float()), R(sd["multi_modal_projector.linear_2.bias"].float())))


def phi3_prefill_logits(embeds: T, lengths: Sequence[int], sd: Dict[str, T], layers: int, heads: int, kv_heads: int,
                        rms_eps: float = 1e-5, theta: float = 10000.0, lowp=None) -> T:
    """embeds (B,S,H) right-padded f32 -> logits (B,vocab) at position lengths[b]-1."""
    m = "language_model.model."
    B, S, H = embeds.shape
    hd = H // heads
    R = _rounder(lowp)
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None]
    cos, sin = R(torch.cat([ang.cos(), ang.cos()], -1)), R(torch.cat([ang.sin(), ang.sin()], -1))    # HF casts cos / sin to the activations' dtype

    def rms(x, w):           # HF Phi3RMSNorm: weight * x_hat.to(input_dtype)
        return R(R(w.float()) * R(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + rms_eps)))

    def rot(x):   # (B,h,S,hd): (x * cos) + (rotate_half(x) * sin), every product stored
        x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
        return R(R(x * cos) + R(torch.cat([-x2, x1], -1) * sin))

    causal = torch.ones(S, S, dtype=torch.bool).tril()
    x = R(embeds.float())
    for i in range(layers):
        p = f"{m}layers.{i}."
        h = rms(x, sd[p + "input_layernorm.weight"])
        qkv = R(F.linear(h, R(sd[p + "self_attn.qkv_proj.weight"].float())))
        q = qkv[..., : heads * hd].view(B, S, heads, hd).transpose(1, 2)
        k = qkv[..., heads * hd: (heads + kv_heads) * hd].view(B, S, kv_heads, hd).transpose(1, 2)
        v = qkv[..., (heads + kv_heads) * hd:].view(B, S, kv_heads, hd).transpose(1, 2)
        q, k = rot(q), rot(k)
        if kv_heads != heads:
            k, v = k.repeat_interleave(heads // kv_heads, 1), v.repeat_interleave(heads // kv_heads, 1)
        att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        att = torch.softmax(att.masked_fill(~causal, float("-inf")), -1)
        a = R(R(att) @ v).transpose(1, 2).reshape(B, S, heads * hd)
        x = R(x + R(F.linear(a, R(sd[p + "self_attn.o_proj.weight"].float()))))
        h = rms(x, sd[p + "post_attention_layernorm.weight"])
        gu = R(F.linear(h, R(sd[p + "mlp.gate_up_proj.weight"].float())))
        g, u = gu.chunk(2, -1)
        x = R(x + R(F.linear(R(u * R(F.silu(g))), R(sd[p + "mlp.down_proj.weight"].float()))))
    last = x[torch.arange(B), torch.as_tensor(lengths) - 1]
    return R(F.linear(rms(last, sd[m + "norm.weight"]), R(sd["language_model.lm_head.weight"].float())))


def phi3_greedy_decode(embeds: T, lengths: Sequence[int], sd: Dict[str, T], layers: int, heads: int, kv_heads: int, max_new_tokens: int,
                       end_id=None, rms_eps: float = 1e-5, theta: float = 10000.0, forced=None, lowp=None):
    """Greedy generation by DEFINITION -- `llava.generate(inputs_embeds=..., max_new_tokens=20, do_sample=False)` at VLN-POL:463
    (transformers GenerationMixin greedy search, an un-vendored dependency): at every step the whole prefix is run again through
    `phi3_prefill_logits` (no KV cache), the argmax token's embedding is appended to that sequence.  Small cases only.
    Returns (tokens per sequence up to and including end_id, logits (steps,B,vocab)).  `forced` (steps x B) replaces the argmax."""
    emb_w = _rounder(lowp)(sd["language_model.model.embed_tokens.weight"].float())
    B, S, H = embeds.shape
    emb = torch.zeros(B, S + max_new_tokens, H)
    emb[:, :S] = embeds.float()
    lens = [int(n) for n in lengths]
    gen, done, steps = [[] for _ in range(B)], [False] * B, []
    for i in range(max_new_tokens):
        lo = phi3_prefill_logits(emb[:, : max(lens)], lens, sd, layers, heads, kv_heads, rms_eps, theta, lowp)
        steps.append(lo)
        nxt = lo.argmax(-1) if forced is None else torch.as_tensor(forced[i])
        for b in range(B):
            if not done[b]:
                gen[b].append(int(nxt[b]))
                done[b] = end_id is not None and int(nxt[b]) == end_id
        if i == max_new_tokens - 1 or all(done):
            break
        for b in range(B):                                   # every sequence advances (finished ones are simply ignored afterwards)
            emb[b, lens[b]] = emb_w[int(nxt[b])]
            lens[b] += 1
    return gen, torch.stack(steps)


def prefix_tokens(info6: T, ifts: T, irel: T, zfts: T, zrel: T, sd: Dict[str, T], lowp=None):
    """VLN-POL:432-435.  info6 (N,576,6) = [x,y,z,sin d,cos d,scale].
    lowp: the product evaluates the SECOND (LM-width x LM-width) layer of `patch_position_embedding`, `instance_projector` and
    `zone_projector` as a 16-bit GEMM in the LM's dtype (their outputs become LM-dtype tokens; the reference runs all of these
    MLPs under fp16 autocast, VLN-TR:385): 16-bit operands and a 16-bit store for that layer, float32 before it."""
    f = {k: v.float() for k, v in sd.items() if k.split(".")[0] in ("patch_position_embedding", "instance_position_embedding",
                                                                     "zone_position_embedding", "instance_projector", "zone_projector")}
    R = _rounder(lowp)

    def mlp2(x, name):
        h = F.gelu(NN.layer_norm(NN.linear(x, f, name + ".0"), f, name + ".1", 1e-5))
        return R(F.linear(R(h), R(f[name + ".3.weight"]), R(f[name + ".3.bias"])))

    patch = mlp2(info6, "patch_position_embedding")
    inst = mlp2(torch.cat([ifts, NN.mlp_ln_gelu(irel, f, "instance_position_embedding")], -1), "instance_projector")
    zone = mlp2(torch.cat([zfts, NN.mlp_ln_gelu(zrel, f, "zone_position_embedding")], -1), "zone_projector")
    return patch, inst, zone
