"""Differentiable device operators of the pre-training step: `torch.autograd.Function`s whose forward AND backward are
HIP kernels of libdynam3d_hip.so -- the float32 Linear (three launches of the float32 MFMA GEMM), LayerNorm [+ GELU], the packed
variable-length set attention, the alpha compositing of the novel-view renderer.  On CPU tensors (the GPU-less test suite) every
operator evaluates the SAME expression with PyTorch, so the host logic around them runs anywhere; on a CUDA tensor there is no fallback.

Reference: the modules of PRE-FF:134-161 (`nn.Linear`, `nn.LayerNorm`, `nn.GELU`, `nn.TransformerEncoderLayer`) and `raw2feature`
(PRE-FF:446-474) under `loss.backward()` (PRE-TR:512)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from ._lib import f32, i32, i64, vp

_lib.register("d3d_layer_norm_bwd_f32", [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, f32, i32, vp])
_lib.register("d3d_layer_norm_bwd_rows_per_block", [])
_lib.register("d3d_gelu_f32", [vp, vp, vp, i64, vp])
_lib.register("d3d_set_attention_bwd", [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp])
_lib.register("d3d_composite_bwd", [vp, i64, vp, i64, vp, vp, vp, i32, i32, i32, vp, vp, vp])


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return _lib.current_stream_ptr()


def _f32ops():
    from .f32_ops import F32Ops
    global _F32
    try:
        return _F32
    except NameError:
        _F32 = F32Ops()
        return _F32


# ---------------------------------------------------------------------------------------------------------------------------------
class _LayerNormF32(torch.autograd.Function):
    """y = [gelu](LayerNorm(x) * w + b), float32: forward d3d_layer_norm_f32, backward d3d_layer_norm_bwd_f32."""

    @staticmethod
    def forward(ctx, x, w, b, eps, gelu):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2, w, b)
        ctx.eps, ctx.gelu, ctx.shape = float(eps), bool(gelu), x.shape
        return _f32ops().layer_norm(x2, w.detach(), b.detach(), eps, gelu=gelu).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, b = ctx.saved_tensors
        lib = _lib.load()
        rows, D = x2.shape
        dy2 = dy.reshape(rows, D).contiguous()
        dx = torch.empty_like(x2)
        rpb = int(lib.d3d_layer_norm_bwd_rows_per_block())
        nb = (rows + rpb - 1) // rpb
        dwp = torch.empty((nb, D), dtype=torch.float32, device=x2.device)
        dbp = torch.empty((nb, D), dtype=torch.float32, device=x2.device)
        _lib.check(lib.d3d_layer_norm_bwd_f32(_p(x2), _p(w.detach().contiguous()), _p(b.detach().contiguous()), _p(dy2), _p(dx), _p(dwp), _p(dbp), rows, D,
                                              x2.stride(0), dy2.stride(0), dx.stride(0), ctx.eps, 1 if ctx.gelu else 0, _stream()))
        return dx.view(ctx.shape), dwp.sum(0), dbp.sum(0), None, None


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, gelu: bool = False) -> torch.Tensor:
    if not x.is_cuda:
        y = F.layer_norm(x, (x.shape[-1],), w, b, eps)
        return F.gelu(y) if gelu else y
    if x.shape[0] == 0:
        return x
    return _LayerNormF32.apply(x, w, b, eps, gelu)


class _GeluF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        ctx.save_for_backward(z)
        out = torch.empty_like(z)
        _lib.check(_lib.load().d3d_gelu_f32(_p(z), None, _p(out), z.numel(), _stream()))
        return out

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        dz = torch.empty_like(z)
        _lib.check(_lib.load().d3d_gelu_f32(_p(z), _p(dy.contiguous()), _p(dz), z.numel(), _stream()))
        return dz


def gelu(z: torch.Tensor) -> torch.Tensor:
    if not z.is_cuda or z.numel() % 4 or z.numel() == 0:
        return F.gelu(z)
    return _GeluF32.apply(z)


# ---------------------------------------------------------------------------------------------------------------------------------
class _SetAttention(torch.autograd.Function):
    """Packed variable-length self-attention inside token sets (d3d_set_attention / d3d_set_attention_bwd): qkv (T, 3*H*64) float32,
    set g = rows [set_off[g], set_off[g + 1]); `q_rows` = 1 evaluates the first row of every set only (the other rows' output is zero)."""

    @staticmethod
    def forward(ctx, qkv, set_off, n_sets, n_heads, max_len, q_rows):
        from .hip_dense import HipDense
        qkv = qkv.contiguous()
        out = HipDense().set_attention(qkv, set_off, n_sets, n_heads, max_len, q_rows=q_rows)
        ctx.save_for_backward(qkv, out, set_off)
        ctx.args = (int(n_sets), int(n_heads), int(max_len), int(q_rows))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, set_off = ctx.saved_tensors
        n_sets, H, max_len, q_rows = ctx.args
        lib = _lib.load()
        T = qkv.shape[0]
        dqkv = torch.zeros_like(qkv) if q_rows > 0 else torch.empty_like(qkv)
        lse = torch.empty((T, H), dtype=torch.float32, device=qkv.device)
        dsum = torch.empty((T, H), dtype=torch.float32, device=qkv.device)
        _lib.check(lib.d3d_set_attention_bwd(_p(qkv), _p(out), _p(dout.contiguous()), _p(set_off), n_sets, H, max_len, q_rows, _p(dqkv), _p(lse), _p(dsum),
                                             _stream()))
        return dqkv, None, None, None, None, None


def set_attention(qkv: torch.Tensor, set_off: torch.Tensor, lens, n_heads: int, q_rows: int = 0) -> torch.Tensor:
    """qkv (T, 3*H*64); lens (host) the set lengths; -> (T, H*64).  CPU: per-set softmax attention in PyTorch."""
    lens = np.asarray(lens, np.int64)
    if qkv.is_cuda:
        return _SetAttention.apply(qkv, set_off, len(lens), n_heads, int(lens.max()), q_rows)
    T, H = qkv.shape[0], n_heads
    out = torch.zeros((T, H * 64), dtype=qkv.dtype)
    o = 0
    outs = []
    for n in lens.tolist():
        blk = qkv[o:o + n].view(n, 3, H, 64)
        q, k, v = (blk[:, j].transpose(0, 1) for j in range(3))                  # (H, n, 64)
        nq = n if q_rows <= 0 else min(q_rows, n)
        a = F.scaled_dot_product_attention(q[:, :nq][None], k[None], v[None])[0].transpose(0, 1).reshape(nq, H * 64)
        outs.append(torch.cat([a, torch.zeros((n - nq, H * 64), dtype=qkv.dtype)], 0) if nq < n else a)
        o += n
    return torch.cat(outs, 0) if outs else out


# ---------------------------------------------------------------------------------------------------------------------------------
class _Composite(torch.autograd.Function):
    """raw2feature (PRE-FF:446-474): (n_rays * S, 768) sample features + (n_rays * S) densities -> (n_rays, 768) unit-norm feature map.
    Inputs are float32 tensors holding fp16-representable values (the tcnn networks' outputs); the kernels read them as fp16."""

    @staticmethod
    def forward(ctx, feat, dens, rel_dist16, topk, n_samples):
        from . import render  # noqa: F401  (registers d3d_composite)
        lib = _lib.load()
        f16, d16 = feat.detach().to(torch.float16).contiguous(), dens.detach().to(torch.float16).contiguous()
        n_rays, S = topk.shape
        fmap = torch.empty((n_rays, f16.shape[1]), dtype=torch.float32, device=feat.device)
        depth = torch.empty((n_rays,), dtype=torch.float32, device=feat.device)
        _lib.check(lib.d3d_composite(_p(f16), f16.stride(0), _p(d16), 1, _p(rel_dist16), _p(topk), n_rays, int(n_samples), S, _p(fmap), _p(depth), _stream()))
        ctx.save_for_backward(f16, d16, rel_dist16, topk)
        ctx.n_samples = int(n_samples)
        ctx.mark_non_differentiable(depth)
        return fmap, depth

    @staticmethod
    def backward(ctx, gout, _gdepth):
        f16, d16, rel_dist16, topk = ctx.saved_tensors
        lib = _lib.load()
        n_rays, S = topk.shape
        dfeat = torch.empty((n_rays * S, f16.shape[1]), dtype=torch.float32, device=f16.device)
        ddens = torch.empty((n_rays * S,), dtype=torch.float32, device=f16.device)
        _lib.check(lib.d3d_composite_bwd(_p(f16), f16.stride(0), _p(d16), 1, _p(rel_dist16), _p(topk), _p(gout.contiguous()), n_rays, ctx.n_samples, S,
                                         _p(dfeat), _p(ddens), _stream()))
        return dfeat, ddens, None, None, None


def composite_reference(feat: torch.Tensor, dens: torch.Tensor, rel_dist: torch.Tensor, topk: torch.Tensor, n_samples: int):
    """The same expression in PyTorch (any device / dtype): the CPU path of `composite`, and the autograd reference the kernels are tested
    against.  feat (n*S, F), dens (n*S), rel_dist (N,), topk (n, S) sample indices."""
    n, S = topk.shape
    tk = topk.long()
    sp = F.softplus(dens.view(n, S))
    rd = rel_dist.to(feat.dtype)
    nxt = torch.cat([rd[1:], rd[-1:]])                                            # (the last bin's width is replaced below)
    dist = torch.where(tk + 1 < n_samples, (nxt[tk] - rd[tk]).abs(), torch.full_like(sp, 1e10))
    alpha = 1.0 - torch.exp(-torch.relu(sp) * dist)
    before = (tk[:, None, :] < tk[:, :, None]).to(feat.dtype)                     # [ray, t, u]: sample u lies in front of sample t
    T = torch.exp((torch.log((1.0 - alpha) + 1e-10)[:, None, :] * before).sum(-1))
    w = alpha * T
    acc = (w[..., None] * feat.view(n, S, -1)).sum(1)
    fmap = acc / torch.clamp(torch.linalg.norm(acc, dim=-1, keepdim=True), min=1e-7)
    depth = (w * rd[tk]).sum(-1) / torch.clamp(w.sum(-1), min=1e-7)
    return fmap, depth


def composite(feat: torch.Tensor, dens: torch.Tensor, rel_dist16: torch.Tensor, topk: torch.Tensor, n_samples: int):
    if feat.is_cuda:
        return _Composite.apply(feat, dens, rel_dist16, topk, n_samples)
    return composite_reference(feat, dens, rel_dist16, topk, n_samples)


class RoundFp16(torch.autograd.Function):
    """x -> fp16 -> float32 with an identity gradient: a 16-bit STORE inside a float32 training graph (a real `.half().float()` would
    also cast the gradient to fp16 on the way back and flush the 1e-6-sized gradients of mean-reduced losses)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.float16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g
