"""Dense primitive dispatch for the towers.  Each primitive names the HIP kernel that implements it on
gfx950 (csrc/gemm_kernels.hip, attn_kernels.hip, dense_kernels.hip); primitives that do not have a
hand-written kernel yet run on PyTorch-ROCm's MFMA libraries (hipBLASLt GEMM / SDPA flash attention),
which the north_star allows for plain library contractions.  `BACKEND` records the choice per primitive
so the benchmark can report exactly what ran.

Shapes: activations are row-major (tokens, features); weights are [out, in] like nn.Linear.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

BACKEND = {"linear": "torch/hipBLASLt", "attention": "torch/SDPA", "layer_norm": "torch", "rms_norm": "torch",
           "rope": "torch", "swiglu": "torch", "resize_normalize": "torch"}

_hip = None  # set by enable_hip_kernels()


def enable_hip_kernels(which: Sequence[str] = ("all",)):
    """Switch primitives to the hand-written HIP kernels (requires libdynam3d_hip.so + a GPU)."""
    global _hip
    from . import hip_dense
    _hip = hip_dense.HipDense()
    names = list(BACKEND) if "all" in which else list(which)
    for n in names:
        if _hip.has(n):
            BACKEND[n] = "hip"
    return dict(BACKEND)


def _act(y, act):
    if act is None:
        return y
    if act == "quick_gelu":               # clip/model.py:162-164
        return y * torch.sigmoid(1.702 * y)
    if act == "gelu":
        return F.gelu(y)
    raise ValueError(act)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: Optional[str] = None) -> torch.Tensor:
    if BACKEND["linear"] == "hip" and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16):
        return _hip.linear(x, w, b, act)
    return _act(F.linear(x, w, b), act)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """float32 statistics and affine, result in x.dtype (clip/model.py:153-159)."""
    if BACKEND["layer_norm"] == "hip" and x.is_cuda:
        return _hip.layer_norm(x, w, b, eps)
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).to(x.dtype)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    if BACKEND["rms_norm"] == "hip" and x.is_cuda:
        return _hip.rms_norm(x, w, eps)
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w).to(x.dtype)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool) -> torch.Tensor:
    """q,k,v (B,L,H,hd) (any strides) -> (B,L,H,hd) contiguous; softmax scale 1/sqrt(hd)."""
    if BACKEND["attention"] == "hip" and q.is_cuda:
        return _hip.attention(q, k, v, causal)
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal)
    return o.transpose(1, 2).contiguous()


def rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Half-split rotary embedding (HF `rotate_half`): pairs (i, i+hd/2).  x (B,S,H,hd); cos/sin (S,hd/2) f32."""
    if BACKEND["rope"] == "hip" and x.is_cuda:
        return _hip.rope(x, cos, sin)
    hd = x.shape[-1]
    x1, x2 = x[..., : hd // 2].float(), x[..., hd // 2:].float()
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(x.dtype)


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    """Phi-3 MLP: gate, up = chunk(gate_up, 2); up * silu(gate)."""
    if BACKEND["swiglu"] == "hip" and gu.is_cuda:
        return _hip.swiglu(gu)
    g, u = gu.chunk(2, dim=-1)
    return (u.float() * F.silu(g.float())).to(gu.dtype)


def resize_normalize(rgb_u8: torch.Tensor, size: int, mean, std) -> torch.Tensor:
    if BACKEND["resize_normalize"] == "hip" and rgb_u8.is_cuda:
        return _hip.resize_normalize(rgb_u8, size, mean, std)
    x = rgb_u8.permute(0, 3, 1, 2).float()
    if x.shape[-1] != size or x.shape[-2] != size:
        x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=False)
        x = x.round().clamp(0, 255)                   # torchvision casts the interpolated image back to uint8
    x = x / 255.0
    m = torch.tensor(mean, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    return (x - m) / s
